#!/bin/bash
# Measurement: the bf16 x 3 weight gradient (conv_wgrad_mma_kernel): backward parity on the GPU, the training step's kernel table, e2e_train
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_frow_backward.py tests/test_training_step.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/train_profile.py 2>&1 | grep -v amdgpu | head -12
timeout 300 python tools/train_profile.py 2>&1 | tail -1
timeout 300 python bench.py --config e2e_train --steps 5 --warmup 2 2>&1 | tail -1 | cut -c1-400
