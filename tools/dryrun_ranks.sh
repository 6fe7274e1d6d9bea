#!/bin/bash
# Multi-PROCESS first-run check of the N > 1 bench path on ONE GPU (all ranks share GPU 0, gloo instead of RCCL, a small workload): the
# launcher as the driver uses it, the topology, the golden self-check through every rank's shard, the grouped-exchange probe (SPECS entry
# "sp:8:fail" = rank 1 reports a stalled probe: every rank must fall back to one exchange per attention and the run must still finish),
# the comm microbench, the JSON line.  NOT a measurement.  (Under gloo bench.py runs one exchange per attention by itself; with 2-rank
# groups gloo's device staging stalls on the grouped exchange some blocks into a forward -- profiles/r03/dryrun_ranks_small.txt, and
# again in round 4 with FW_SP_EXCHANGE_GROUPS=2 forced -- which no short probe can promise to catch; RCCL runs a communicator's
# collectives in issue order, where the same issue order on every rank is sufficient.)
# Round 6: a fourth field selects BASELINE config 5's arithmetic -- "sp:4::fp8" / "tp:4::fp8" run `--model wan22 --precision fp8
# --fp8-attention` (fp8 linears + fp8 attention under the partition, golden self-check of the fp8 engines included, the `alt` block
# on the OTHER partition with the same options).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${TAG:-dry}; mkdir -p $O; cd $R
export FW_BENCH_DEVICE=0 FW_DIST_BACKEND=gloo FW_SP_PROBE_TIMEOUT_S=${FW_SP_PROBE_TIMEOUT_S:-15} FW_BENCH_WATCHDOG_S=${FW_BENCH_WATCHDOG_S:-300}
for spec in ${SPECS:-sp:4 tp:4 sp:8:grouped sp:8:fail}; do
  IFS=: read mode n how prec <<< "$spec"
  extra=""; [ "$prec" = "fp8" ] && extra="--model wan22 --precision fp8 --fp8-attention"
  unset FW_SP_EXCHANGE_GROUPS FW_SP_PROBE_FORCE_FAIL
  [ "$how" = "grouped" ] && export FW_SP_EXCHANGE_GROUPS=2
  [ "$how" = "fail" ] && export FW_SP_EXCHANGE_GROUPS=2 FW_SP_PROBE_FORCE_FAIL=1
  FW_PARALLEL=$mode timeout ${T_SPEC:-600} python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) \
    bench.py --gpus $n --steps 2 --warmup 1 --no-cpu-baseline --layers 18 --frames 17 --height 128 --width 128 $extra > $O/dryrun_${TAG}_${mode}_$n$how$prec.log 2>&1
  echo "== $mode n=$n $how $prec exit $?" | tee -a $O/dryrun_$TAG.txt
  grep '^{' $O/dryrun_${TAG}_${mode}_$n$how$prec.log | tail -n 1 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    c = d.get('comm') or {}
    print(json.dumps({k: d.get(k) for k in ('value', 'n_gpus', 'ms_per_step', 'dtype', 'golden_check', 'watchdog_s', 'error')}))
    print(json.dumps({'parallelism': d['config'].get('parallelism'), 'exchange_groups': c.get('exchange_groups'), 'tp_reduce_dtype': c.get('tp_reduce_dtype'), 'microbench': c.get('microbench')}))
    a = d.get('alt')
    print(json.dumps({'alt': None if a is None else {k: a.get(k) for k in ('parallelism', 'value', 'ms_per_step', 'steps', 'golden_check', 'engine_build_s', 'error')}}))
" | tee -a $O/dryrun_$TAG.txt
  grep -i "error\|Traceback\|Timeout" $O/dryrun_${TAG}_${mode}_$n$how$prec.log | head -5 | tee -a $O/dryrun_$TAG.txt
done
