#!/bin/bash
# Lists every kernel of csrc/*.hip whose gfx950 code uses scratch memory (register spills, or the by-value kernel argument
# copied to scratch because something indexes it at run time) with its VGPR count:   bash tools/check_scratch.sh [file.hip ...]
# (round 4: a second pick loop over pg_dst_t fields put 1.3 KB of scratch per lane into the fp32 headline kernel unnoticed)
cd "$(dirname "$0")/.."
C=pose-transfer_amd/csrc
FILES=${@:-$(ls $C/*.hip)}
T=$(mktemp -d)
for f in $FILES; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -Iinclude -I$C \
      -Rpass-analysis=kernel-resource-usage -c $f -o $T/$(basename $f).o 2> $T/$(basename $f).log ) &
done
wait
for f in $FILES; do
  grep -E "Function Name: |ScratchSize|VGPRs:" $T/$(basename $f).log | sed 's/.*remark: *//' | paste - - - | grep -v "lane\]: 0" |
    sed 's/\[-Rpass[^]]*\]//g; s/Function Name: //' | while read -r name rest; do echo "$(basename $f): $(echo $name | c++filt | cut -c1-110) | $rest"; done
done
rm -rf $T
