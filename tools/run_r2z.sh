for rep in 1 2; do
for v in base casc7 andshift; do
export LD_LIBRARY_PATH=$PWD/variants/$v
echo "$v 1280 trig: $(timeout -k 5 120 ./tools/k2_microbench 4000 8 0 1280 1024 | tail -1 | cut -c84-140)"
echo "$v 1280 store: $(timeout -k 5 120 ./tools/k2_microbench 4000 8 1 1280 1024 | tail -1 | cut -c84-140)"
echo "$v 1680 trig: $(timeout -k 5 120 ./tools/k2_microbench 3000 8 0 1680 1050 | tail -1 | cut -c84-140)"
echo "$v 1680 store: $(timeout -k 5 120 ./tools/k2_microbench 3000 8 1 1680 1050 | tail -1 | cut -c84-140)"
done; done
