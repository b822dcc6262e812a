#!/bin/bash
# kernel timeline of the last train() of one model kind: tools/trace_mode.sh "indexed=1" | "doskipgrams_exhaustive=1" | "indexed=1,doskipgrams=1"
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/trace_mode; rm -rf $O; mkdir -p $O
cat > $O/run.py <<PY
import sys
sys.path.insert(0, '$GRAFT_REPO_ROOT/colibri-core_amd/pyhost')
from colibri_amd import capi, synth
payload = synth.zipf_corpus(100_000_000, 1_000_000, 44, header=False)
with capi.Context(0) as c:
    c.upload(payload)
    for rep in range(2):
        st = c.train(maxlength=5, mintokens=2, ${1})
    print('train ms', st.train_ms, st.npatterns, st.nrefs)
PY
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $O/run.py > $O/log.txt 2>&1
tail -1 $O/log.txt
python $GRAFT_REPO_ROOT/tools/trace_step.py $(find $O -name "*kernel_trace.csv" | head -1) uni_head_kernel
