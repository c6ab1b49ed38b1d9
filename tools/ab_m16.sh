set -e
mkdir -p gpurun_out/r02_m16
export FUSG_LIB=$PWD/future_urban_scene_generation_amd/libfusg_m16.so FUSG_HALO_M16=1
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x > gpurun_out/r02_m16/ops_m16.log 2>&1 || true
tail -5 gpurun_out/r02_m16/ops_m16.log
for rep in 1 2; do
  unset FUSG_LIB FUSG_HALO_M16
  timeout -k 10 300 python tools/halo_exp.py > gpurun_out/r02_m16/base_$rep.txt 2>&1
  export FUSG_LIB=$PWD/future_urban_scene_generation_amd/libfusg_m16.so FUSG_HALO_M16=1
  timeout -k 10 300 python tools/halo_exp.py > gpurun_out/r02_m16/m16_$rep.txt 2>&1
done
paste -d'\n' gpurun_out/r02_m16/base_1.txt gpurun_out/r02_m16/m16_1.txt gpurun_out/r02_m16/base_2.txt gpurun_out/r02_m16/m16_2.txt | grep -v amdgpu.ids
