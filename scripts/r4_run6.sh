#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout -s KILL 500 python -m pytest tests/test_gpu_model.py tests/test_gpu_ddp.py -q -m gpu -x -k "split_stream or graphed_train or world1 or ranks or train_step_matches or graphed_decode or eval_after" 2>&1 | tail -30 ) > $O/r4_6_tests.log
( timeout -s KILL 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r4_6_bench.json ) 2> $O/r4_6_bench.err
cat $O/r4_6_tests.log; tail -c 400 $O/r4_6_bench.err; python - <<'P'
import json
txt=open('gpurun_out/r4_6_bench.json').read()
line=[l for l in txt.splitlines() if l.startswith('{')]
if line:
    d=json.loads(line[-1])
    print(d['value'], d['ms_per_step'], d['config']['hipgraph'], d['config'].get('graph_form'), d['config']['launch_probe'], d['config']['host_enqueue_ms_per_step'])
    for k,v in d['configs'].items():
        if k=='ddp_world1':
            for kk,vv in v.items():
                if isinstance(vv,dict): print('  ddp',kk,{m:(vv[m].get('ms_per_step'),vv[m].get('host_enqueue_ms_per_step'),vv[m].get('vs_no_group'),vv[m].get('error')) for m in ('eager','hipgraph')}, vv.get('faster'))
        elif 'error' in v: print(k, v)
        elif k=='synth_rtf': print(k, v['value'], v['config']['model_ms'], v['config']['vocoder_ms'])
        else: print(k, v['value'], v['ms_per_step'], v['config']['hipgraph'], v['config']['launch_probe'], v['config']['host_enqueue_ms_per_step'])
    print(d.get('input_pipeline'))
P
