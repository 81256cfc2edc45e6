cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in "1 1" "1 0" "0 0"; do set -- $v
  USIP_NARROW_BWD=$1 USIP_NARROW_RED=$2 python bench.py --no-cpu-baseline --no-kernel-timing --no-kernel-leg --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused=$1 red=$2  %.3f ms  p10 %.3f p90 %.3f' % (d['ms_per_step'], d['step_ms_rank0']['p10'], d['step_ms_rank0']['p90']))"
done; done
