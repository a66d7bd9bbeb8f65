cd $GRAFT_REPO_ROOT
echo "== cli"; python - <<'PY'
import subprocess, time, os, sys, shutil, tempfile
scratch = tempfile.mkdtemp(prefix='maray_cli_')
root=os.environ['GRAFT_REPO_ROOT']
cli=os.path.join(root,'maray_amd','maray')
if not os.path.exists(cli):
    cli=None
    for c in ('maray_amd/csrc/maray','maray_amd/bin/maray'):
        if os.path.exists(os.path.join(root,c)): cli=os.path.join(root,c)
print('cli', cli)
if cli:
    for tag, env, args in (('auto nothing cached', {'MARAY_CACHE_DIR': scratch + '/a', 'AMD_COMGR_CACHE_DIR': scratch + '/ca'}, ['--backend','auto']), ('jit cold', {'MARAY_CACHE_DIR': scratch + '/b', 'AMD_COMGR_CACHE_DIR': scratch + '/cb'}, ['--backend','jit']), ('jit warm', {'MARAY_CACHE_DIR': scratch + '/b', 'AMD_COMGR_CACHE_DIR': scratch + '/cb'}, ['--backend','jit'])):
        e=dict(os.environ); e.update(env)
        t=time.perf_counter(); r=subprocess.run([cli,'-i',os.path.join(root,'tests/golden/chess.maray'),'-o','/tmp/chess_cli.png']+args, env=e, capture_output=True, text=True); dt=time.perf_counter()-t
        print(tag, round(dt,2), 's rc', r.returncode, r.stderr[-200:])
PY
