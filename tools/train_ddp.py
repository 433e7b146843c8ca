#!/usr/bin/env python
"""Data-parallel training step of the interaction hot path on synthetic nuScenes-shaped batches (BASELINE.json configs[2]
at 1 GPU, configs[3] at N GPUs) - a thin front end of `bench.py --mode train` (deepinteraction_amd/train_step.py: forward
in train() mode, head loss with the Hungarian assignment on the host, backward through the HIP kernels, gradient buckets
all-reduced over RCCL from backward hooks (`parallel.GradientReducer`), grad clip, AdamW).

    python tools/train_ddp.py --steps 10                    (1 GPU)
    python tools/train_ddp.py --gpus N --steps 10           (N GPUs: launches its own ranks, one per GPU)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, 'bench.py'), '--mode', 'train'] + sys.argv[1:]))
