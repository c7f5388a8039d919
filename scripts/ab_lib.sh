q() { python bench.py --no-learn-loop --no-cpu-baseline --repeats 3 $1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
for i in 1 2; do
echo -n "base sac: "; q
echo -n "kp   sac: "; GRL_LIBRARY=deep-rl-grasping_amd/grasp_rl/libgrl_kp.so q
done
echo -n "base bdq: "; q "--workload bdq_per"
echo -n "kp   bdq: "; GRL_LIBRARY=deep-rl-grasping_amd/grasp_rl/libgrl_kp.so q "--workload bdq_per"
