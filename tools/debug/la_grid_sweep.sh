for g in 512 480 448 424 400 352; do echo "grid $g"; DI_LA_GRID=$g timeout 100 python tools/la_floor.py 2>&1 | grep "generation"; done
