#!/bin/bash
for b in 1024 2048 4096 8192; do echo "blocks $b: $(DI_I2P_BLOCKS=$b python tools/i2p_bench.py 2>/dev/null | grep i2p_attention)"; echo "blocks $b row-major: $(DI_I2P_NO_ORDER=1 DI_I2P_BLOCKS=$b python tools/i2p_bench.py 2>/dev/null | grep i2p_attention)"; done
