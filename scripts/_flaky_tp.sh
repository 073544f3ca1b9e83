cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in $(seq 1 24); do
    TEAL_TP_WORKER_DIGESTS=1 TEAL_TP_BACKEND=gloo OMP_NUM_THREADS=8 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 2000)) tests/tp_gpu_worker.py 7B fp16 2>/dev/null | grep 'RANKDIGEST' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l[11:])
    for lab in ('dense','sparse'):
        g=d['digest'][lab]; e=g['early']
        print($i, 'rank', d['rank'], lab, 'decode_full', g['decode_full'], 'early', e['decode_full'], 'SAME' if g['decode_full']==e['decode_full'] else 'CHANGED-LATER', '| kc', g['kc_full'], e['kc_full'], 'SAME' if g['kc_full']==e['kc_full'] else 'CHANGED-LATER', hex(e['ptr_log_f']), hex(e['ptr_kc_f']))"
done
