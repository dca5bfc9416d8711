python tools/e2e_files.py /tmp/e2e 4 > /dev/null 2>&1
g++ -O2 -std=c++17 -Iuniversal-volumetric_amd/host tools/ingest_bench.cpp universal-volumetric_amd/host/uvol_host.cpp -o /tmp/ingest_bench -lz -ldl -lpthread
lscpu | grep -i "model name\|socket\|core(s)\|thread(s)\|numa node(s)" | head -6
for n in 1 8 32 64 128 256; do /tmp/ingest_bench /tmp/e2e/OBJ/frame_00000.obj /tmp/e2e/PNG/export_00000.png $n 4; done
