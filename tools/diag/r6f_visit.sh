mkdir -p gpurun_out/r6f
timeout 1500 python -m pytest tests/test_gpu_p2p_options.py tests/test_gpu_train_step.py tests/test_gpu_backward.py tests/test_gpu_bf16.py tests/test_gpu_autograd.py -q -m gpu > gpurun_out/r6f/pytest.log 2>&1
tail -5 gpurun_out/r6f/pytest.log
(python tools/bf16_ab.py --train --steps 8; python tools/bf16_ab.py --depth 50 --size 640 --batch 64 --train --steps 8; CPR_MIXED_DGRAD_1X1=fp32 python tools/bf16_ab.py --train --steps 8 --rounds 2; CPR_BF16_NT_PP=0 python tools/bf16_ab.py --train --steps 8 --rounds 2) > gpurun_out/r6f/train_ab.txt 2>&1
DECOMPOSE=1 timeout 600 python tests/report_mixed_precision_grads.py > gpurun_out/r6f/mixed_decompose.txt 2>&1
