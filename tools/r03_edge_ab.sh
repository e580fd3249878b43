echo "== product"; python3 tools/edge_bench.py 2>/dev/null | cut -c1-200
for so in scratch/libvg_gp*.so; do echo "== $so"; VG_LIB_PATH=$PWD/$so python3 tools/edge_bench.py 2>/dev/null | cut -c1-200; done
