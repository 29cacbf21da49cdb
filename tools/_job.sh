export TMPDIR=/tmp
bash tools/gpu_full.sh round6_final3 both
bash tools/gpu_profile.sh round6_prof3
mkdir -p gpurun_out/round6_stress
timeout 1500 python tools/stress.py 300 6 > gpurun_out/round6_stress/stress.txt 2>&1; tail -3 gpurun_out/round6_stress/stress.txt; grep -c "pcm16 == f32" gpurun_out/round6_stress/stress.txt; grep -c "CHECK" gpurun_out/round6_stress/stress.txt
