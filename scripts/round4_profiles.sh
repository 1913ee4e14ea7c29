set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p4
bash scripts/profile_round.sh round4_v2 > gpurun_out/p4/profile_round.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/p4/round4_v2_bench.json 2> gpurun_out/p4/bench.err
python bench.py --steps 10 --warmup 3 --landmark-order random --iteration-mix r2 --no-cpu-baseline --single-reps 0 > gpurun_out/p4/round4_v2_r2workload_bench.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --landmark-order random --no-cpu-baseline --single-reps 0 > gpurun_out/p4/round4_v2_random_bench.json 2>/dev/null
python bench.py --workload c2 --steps 5 --warmup 2 > gpurun_out/p4/round4_v2_c2_bench.json 2>/dev/null
python bench.py --workload c3s --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/p4/round4_v2_c3s_bench.json 2>/dev/null
python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/p4/round4_v2_c4_bench.json 2>/dev/null
python bench.py --workload gba --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/p4/round4_v2_gba_bench.json 2>/dev/null
python bench.py --workload pose --steps 5 --warmup 2 > gpurun_out/p4/round4_v2_pose_bench.json 2>/dev/null
ls -la gpurun_out/p4 gpurun_out/prof_round4_v2* 2>/dev/null | tail -30
