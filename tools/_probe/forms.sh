for B in 8 16 32 64 256; do echo "== teddy B=$B"; timeout -k 10 200 python tools/sgm_forms.py 375 450 128 $B 3 2>&1 | tail -6; done
