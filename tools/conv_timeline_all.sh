export PG_DEBUG_CONV_TIMELINE=1
for l in "dec5 fwd" "dec5 dgrad" "dec4 fwd" "dec4 dgrad" "dec3 fwd" "enc1 fwd" "enc1 dgrad" "enc2 fwd" "enc2 dgrad" "enc3 fwd"; do timeout 120 python tools/conv_timeline.py 32 $l 2>&1 | grep -v "amdgpu.ids\|xcd [1-7]"; echo; done
