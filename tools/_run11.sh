cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r04x_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04x_tests.log
tail -3 gpurun_out/r04x_tests.log
for L in "" $GRAFT_REPO_ROOT/ao_marl_amd/variants/libaomarl_hip_nodmaf32.so "" $GRAFT_REPO_ROOT/ao_marl_amd/variants/libaomarl_hip_nodmaf32.so; do
AOMARL_LIB=$L timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-side-configs --no-cpu-baseline --no-whole-episode --timed-only 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('lib [$L] value %.0f  ms/step %.4f no_reset %.4f frame %.4f' % (d['value'], d['ms_per_step'], d['ms_per_step_no_reset'], d['roofline']['avg_launch_ms']))
"
done
