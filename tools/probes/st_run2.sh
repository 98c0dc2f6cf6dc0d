for lib in "" tools/probes/bin/lib_loop.so; do
echo "== lib $lib"
for s in 1025,20,1,2,channels_last,128 1025,83,16,2,channels_last,128 1025,83,128,2,channels_last,128 201,998,128,2,channels_last,40; do
KAPRE_AMD_LIB=$lib python tools/kbench_fb.py 0 shape=$s 2>&1 | tail -1
done
done
