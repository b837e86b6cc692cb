#!/bin/bash
python -m pytest tests/test_dp_gpu.py tests/test_modules_gpu.py tests/test_checkpoint_gpu.py -x -q -m gpu 2>&1 | tail -5
