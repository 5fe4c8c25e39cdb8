# final evidence of round 4, in order of importance (run on the GPU box from the repo root; every step bounded)
set +e
timeout -k 10 420 bash scripts/refresh_profiles_r04.sh resnet50unet_b32_s512_bf16 --net Resnet50Unet > gpurun_out/final_refresh.log 2>&1
timeout -k 10 150 bash scripts/dispatch_profile.sh resnet50unet_b32_s512_bf16 --net Resnet50Unet > gpurun_out/final_disp.log 2>&1
timeout -k 10 900 bash scripts/bench_lines_r04.sh deeplabv3plus fp8 cfg5 cfg5bf16 unetv2 swintupernet segformermod mobilenet cfg1 resnet101 oldwidths > gpurun_out/final_lines.log 2>&1
tail -3 gpurun_out/final_refresh.log | cut -c1-200
tail -2 gpurun_out/final_disp.log
grep -h "tiles/s" gpurun_out/r04_lines/*.txt | cut -c1-60
