#!/bin/bash
cd "$(dirname "$0")/.."
for a in 0 1 8 16; do echo -n "ablate $a: "; python tools/conv_single.py --batch 16 --plain --ablate $a --iters 10 2>/dev/null; done
echo -n "xform+gn: "; python tools/conv_single.py --batch 16 --iters 10 2>/dev/null
