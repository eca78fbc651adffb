#!/bin/bash
# round 5: GPU tests of the new host paths (split training modes, async collectives, audio scale) + world-1 group A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ddp.py tests/test_audio.py -q -m gpu -x -k "split_mode or ddp or nccl or audio or lws or window or train_step" > gpurun_out/r5_tests_b.log 2>&1; echo "tests rc $?" >> gpurun_out/r5_tests_b.log
tail -15 gpurun_out/r5_tests_b.log
timeout 600 python scripts/r5_ddp_world1.py > gpurun_out/r5_ddp_world1_async.txt 2>&1; tail -4 gpurun_out/r5_ddp_world1_async.txt | cut -c1-1500
DV3_COLLECTIVE_STREAM=own timeout 600 python scripts/r5_ddp_world1.py > gpurun_out/r5_ddp_world1_own.txt 2>&1; tail -4 gpurun_out/r5_ddp_world1_own.txt | cut -c1-600
GPU_MAX_HW_QUEUES=8 timeout 600 python scripts/r5_ddp_world1.py > gpurun_out/r5_ddp_world1_async_q8.txt 2>&1; tail -4 gpurun_out/r5_ddp_world1_async_q8.txt | cut -c1-600
