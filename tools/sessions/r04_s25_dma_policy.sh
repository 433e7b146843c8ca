for p in 0 32 64 96 128 160 192 224; do echo "policy bits $p"; DI_RING_DBG=$p timeout 100 python tools/la_bench2.py 24 2>&1 | tail -1; done
