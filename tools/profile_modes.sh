#!/bin/bash
# rocprofv3 --kernel-trace --stats of one train() per model kind of configs 4 / 5 (100 M tokens): exhaustive skipgrams, indexed, indexed + skipgrams.
# usage (on the GPU box, via gpurun): bash tools/profile_modes.sh <tag>   ->  gpurun_out/prof_<tag>_modes/<kind>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_${1:-r02}_modes; rm -rf $O; mkdir -p $O
for spec in "exhaustive_skipgrams:doskipgrams_exhaustive=1" "indexed:indexed=1" "indexed_skipgrams:indexed=1,doskipgrams=1"; do
  kind=${spec%%:*}; kw=${spec#*:}
  cat > $O/run_$kind.py <<PY
import sys
sys.path.insert(0, '$GRAFT_REPO_ROOT/colibri-core_amd/pyhost')
from colibri_amd import capi, synth
payload = synth.zipf_corpus(100_000_000, 1_000_000, 44, header=False)
with capi.Context(0) as c:
    c.upload(payload)
    for rep in range(4):
        st = c.train(maxlength=5, mintokens=2, $kw)
    print('$kind train ms', round(st.train_ms, 2), 'patterns', st.npatterns, 'refs', st.nrefs)
PY
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$kind -o t -- python $O/run_$kind.py > $O/$kind.log 2>&1
  tail -1 $O/$kind.log
  cp $(find $O/$kind -name "*kernel_stats.csv" | head -1) $O/${kind}_kernel_stats.csv
done
ls $O
