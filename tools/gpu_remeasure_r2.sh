cd $GRAFT_REPO_ROOT
export MARAY_CACHE_DIR=/tmp/mc
echo "== strong"; timeout -k 10 300 python bench.py --scaling strong --steps 20 --warmup 5 --cpu-seconds 0 --no-cpu-jit 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step')}, d.get('end_to_end',{}).get('value'), d['roofline']['frac'], d['roofline']['kernel_ms'])"
echo "== launch overhead"; timeout -k 10 120 python tools/exp_launch_overhead.py
echo "== soup 300"; timeout -k 10 300 python tools/bench_soup.py 300 2>&1 | tail -3
echo "== soup 1000"; timeout -k 10 400 python tools/bench_soup.py 1000 2>&1 | tail -3
echo "== soup 300 colours"; timeout -k 10 400 python tools/bench_soup.py 300 colours 2>&1 | tail -3
echo "== store roof"; timeout -k 10 60 python tools/exp_store_roof.py
echo "== cli"; python - <<'PY'
import subprocess, time, os, sys, shutil
root=os.environ['GRAFT_REPO_ROOT']
cli=os.path.join(root,'maray_amd','maray')
if not os.path.exists(cli):
    cli=None
    for c in ('maray_amd/csrc/maray','maray_amd/bin/maray'):
        if os.path.exists(os.path.join(root,c)): cli=os.path.join(root,c)
print('cli', cli)
if cli:
    for tag, env, args in (('auto nothing cached', {'MARAY_CACHE_DIR':'/tmp/mc_cli1'}, ['--backend','auto']), ('jit cold', {'MARAY_CACHE_DIR':'/tmp/mc_cli2'}, ['--backend','jit']), ('jit warm', {'MARAY_CACHE_DIR':'/tmp/mc_cli2'}, ['--backend','jit'])):
        e=dict(os.environ); e.update(env)
        t=time.perf_counter(); r=subprocess.run([cli,'-i',os.path.join(root,'tests/golden/chess.maray'),'-o','/tmp/chess_cli.png']+args, env=e, capture_output=True, text=True); dt=time.perf_counter()-t
        print(tag, round(dt,2), 's rc', r.returncode, r.stderr[-200:])
PY
