#!/bin/bash
# sweep the split-K / M-tile overrides of the direct conv kernel over the microbench layers (one process per setting)
cd "$(dirname "$0")/.."
for tm in 1 2; do for sk in auto 1 2 3 4 6 8 12; do
  echo "== TM=$tm SPLITK=$sk"
  if [ $sk = auto ]; then STEGAN_TM=$tm python scripts/conv_microbench.py 2>/dev/null | cut -c1-36,44-200
  else STEGAN_TM=$tm STEGAN_SPLITK=$sk python scripts/conv_microbench.py 2>/dev/null | cut -c1-36,44-200; fi
done; done
