for v in c3; do for s in 1 4 2 7 8 3 0; do echo -n "$v stop=$s: "; PCREG_LIB=pcreg_amd/variants/$v.so PCREG_DESC_STOP=$s python scripts/desc_dev_bench.py 1000000 100000 | tail -1; done; done
for s in 1 0; do PCREG_DESC_STOP=$s python ab_r2/scripts/desc_ab_stop.py 1000000 100000 | tail -1; done
python scripts/desc_dev_bench.py 1000000 1000000 | tail -1; python scripts/desc_dev_bench.py 1000000 1000000 single | tail -1; python ab_r2/scripts/desc_ab.py 1000000 1000000 | tail -1
