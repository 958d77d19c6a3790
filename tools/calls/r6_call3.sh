# round 6 call 3: cfg 86 (conv_wino4w_kernel) into the table -- every 3x3 s1 shape, with 70 / 80 / 84 timed again in the same session
python tools/retune.py --out gpurun_out/gfx950_r6a.json --match k3x3_s1 --retime 70,80,84,86 2>&1 | tail -40
