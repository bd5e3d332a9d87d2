cd /root/repo
for shp in "--hw 40 --cin 256 --cout 256" "--hw 80 --cin 128 --cout 128" "--hw 20 --cin 512 --cout 512" "--hw 160 --cin 64 --cout 64" "--hw 160 --cin 256 --cout 256 --batch 2"; do for a in "" "--no-wino"; do b="--batch 64"; case "$shp" in *batch*) b="";; esac; timeout 120 python tools/wgrad_single.py $b --iters 5 $shp $a 2>&1 | grep -v amdgpu; done; done
