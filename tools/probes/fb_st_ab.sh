# stereo channels_last rows: k_fb_pw<NC,st> against the MFMA kernel ("fb_variant" 1), tests first (profiles/r06_fb_pw.md section 7)
python -m pytest tests/test_fb_pw.py tests/test_nonfinite.py tests/test_fuzz_gate.py -m gpu -x -q 2>&1 | tail -5
for s in 1025,83,128,2,channels_last,128 1025,83,16,2,channels_last,128 1025,20,1,2,channels_last,128 201,998,128,2,channels_last,40 513,100,64,2,channels_last,96 513,994,64,2,channels_last,80; do
python tools/kbench_fb.py 1 0 shape=$s 2>&1 | tail -5
done
