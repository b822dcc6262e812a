#!/bin/bash
# bi2_ids_kernel: how many windows may be open at a time (COLIBRI_IDS_GRID persistent blocks) — indexed model of the 100 M-token corpus, train ms
for g in 32 64 128 256 1024; do echo "grid $g"; COLIBRI_IDS_GRID=$g python $GRAFT_REPO_ROOT/tools/modes_probe.py 2>&1 | grep -E "^indexed train|^exhaustive"; done
