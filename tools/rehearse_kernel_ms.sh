cd $GRAFT_REPO_ROOT
for nw in "1 0" "2 0" "4 0" "8 0"; do set -- $nw
  echo -n "world=$1 rank=$2: "
  MARAY_BENCH_FAKE_WORLD=$1 MARAY_BENCH_FAKE_RANK=$2 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-cold --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms/step %.4f kernel_ms %.4f value %.0f | %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['config']['workload'][:50]))"
done
