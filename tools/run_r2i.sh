export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
O=gpurun_out/r2i; mkdir -p $O
bash tools/prof_k2.sh r2i_cyc 2000 0 1280 1024 0 1 1 8 > $O/prof_cyc.log 2>&1; python3 tools/summarize_pmc.py gpurun_out/r2i_cyc $O/cyc_summary.json
bash tools/prof_k2.sh r2i_nodisc 2000 0 1280 1024 0 1 1 0 0 > $O/prof_nodisc.log 2>&1; python3 tools/summarize_pmc.py gpurun_out/r2i_nodisc $O/nodisc_summary.json
python3 - <<'PY'
import json
for t in ('cyc','nodisc'):
    d=json.load(open(f'gpurun_out/r2i/{t}_summary.json'))
    print(t, d['micro']['ms_avg'], {k:round(v,3) for k,v in d['wave_cycles_share'].items()}, 'valu/px', round(d['valu_insts_per_pixel_lane'],2))
    print('   ', {k:round(v) for k,v in d['counters_mean'].items()})
PY
