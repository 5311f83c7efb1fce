cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02a/pytest.log
timeout 600 python bench.py > gpurun_out/r02a/bench_survey.json 2> gpurun_out/r02a/bench_survey.err
timeout 600 python bench.py --workload r01 --steps 10 > gpurun_out/r02a/bench_r01.json 2> gpurun_out/r02a/bench_r01.err
timeout 600 python bench.py --model usv_model_guidance_ca1 > gpurun_out/r02a/bench_m1.json 2> gpurun_out/r02a/bench_m1.err
timeout 60 python bench.py --gpus 2 --steps 1 > gpurun_out/r02a/bench_2gpu.out 2>&1; echo "rc $?" >> gpurun_out/r02a/bench_2gpu.out
tail -5 gpurun_out/r02a/pytest.log; cut -c1-600 gpurun_out/r02a/bench_survey.json; tail -3 gpurun_out/r02a/bench_survey.err
