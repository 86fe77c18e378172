mkdir -p gpurun_out/w2
for v in W2SKEL W2SKELNOEPI W2NOEPI; do
  HAPPYPOSE_AMD_LIB=happypose_amd/lib/abl/$v.so HP_PROFILE_LAYERS=1 HP_WINO2=15 timeout 300 python bench.py --lanes 1 --steps 4 --no-cpu-baseline --no-extra-workloads --no-exact-fp32 --render-state single-sample > gpurun_out/w2/abl_$v.log 2>&1
  echo $v $(grep "layer1.1.conv1\|layer2.1.conv1\|layer3.1.conv1\|layer4.1.conv1" gpurun_out/w2/abl_$v.log | awk '{print $(NF-3)}' | tr '\n' ' ')
done
