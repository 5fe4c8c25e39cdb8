# Round-4 bench lines of the workloads that are not the headline (run on the GPU box from the repo root): one JSON line each under
# gpurun_out/r04_lines/, copied to profiles/r04_bench_<tag>.json afterwards.   scripts/bench_lines_r04.sh [tags...]
set -e
mkdir -p gpurun_out/r04_lines
run() { tag=$1; shift; echo "== $tag"; timeout -k 10 420 python3 bench.py "$@" > gpurun_out/r04_lines/$tag.log 2> gpurun_out/r04_lines/$tag.err; tail -1 gpurun_out/r04_lines/$tag.log > gpurun_out/r04_lines/$tag.json; python3 scripts/show_bench.py gpurun_out/r04_lines/$tag.json > gpurun_out/r04_lines/$tag.txt 2>&1 || true; sed -n 1,3p gpurun_out/r04_lines/$tag.txt; }
for t in "$@"; do
  case $t in
    deeplabv3plus) run deeplabv3plus_b32_s512_bf16 --net DeepLabV3Plus ;;
    swintupernet)  run swintupernet_b32_s512_bf16 --net SwinTUperNet ;;
    unetv2)        run unetv2_b32_s512_bf16 --net Unetv2 ;;
    resnet101)     run resnet101_b32_s512_bf16 --net Resnet101 --steps 20 ;;
    segformermod)  run segformermod_b32_s512_bf16 --net SegformerMod ;;
    cfg1)          run cfg1_resnet18unet_b8_s256_nc4_bf16 --net Resnet18Unet --batch 8 --tile 256 --classes 3 ;;
    cfg5)          run cfg5_resnet50unet_b8_s1024_nc21_fp8 --net Resnet50Unet --batch 8 --tile 1024 --classes 20 --precision fp8 ;;
    cfg5bf16)      run cfg5_resnet50unet_b8_s1024_nc21_bf16 --net Resnet50Unet --batch 8 --tile 1024 --classes 20 ;;
    fp8)           run resnet50unet_b32_s512_fp8 --precision fp8 ;;
    swinfp8)       run swintupernet_b32_s512_fp8 --net SwinTUperNet --precision fp8 ;;
    segb3)         run segformermod_b3_b32_s512_bf16 --net SegformerMod --segformer-variant b3 --steps 20 ;;
    unetfp8)       run unetv2_b32_s512_fp8 --net Unetv2 --precision fp8 ;;
    mobilenet)     run mobilenet_b32_s512_bf16 --net MobileNet ;;
    oldwidths)     run resnet50unet_b32_s512_bf16_decoder_256_128_64_64_64 --decoder-channels 256,128,64,64,64 ;;
    nogram)        CVCS_GRAM_BN=0 CVCS_LAZY_HEAD=0 run resnet50unet_b32_s512_bf16_round3_path ;;
  esac
done
