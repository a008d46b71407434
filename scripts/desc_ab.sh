for s in 0 1 4 6 2 3; do PCREG_DESC_STOP=$s python ab_r2/scripts/desc_ab_stop.py 1000000 100000 | tail -1; done
bash scripts/desc_stops.sh double pcreg_amd/variants/c2.so
