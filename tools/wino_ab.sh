cd /root/repo
for e in 0 16384 4096; do echo "== extra lds $e"; timeout 300 python tools/conv_bench.py --batch 64 --extra-lds $e 2>&1 | grep -v amdgpu | grep -E "^\(64, (160|80|40), .*, 1, [12], |conv total" ; done
