#!/bin/bash
cd "$(dirname "$0")/.."
for a in 0 1 2 3 4 7; do echo "ablate $a:"; python tools/conv_single.py --batch 16 --plain --ablate $a --iters 10; done
echo "pipeline 0:"; python tools/conv_single.py --batch 16 --plain --pipeline 0 --iters 10
echo "xform+gn:"; python tools/conv_single.py --batch 16 --iters 10
