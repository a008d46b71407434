for s in 0 6 8 10 12 13 16 19 20 24 27 32; do echo "S=$s"; PCREG_SAD_SPLITS=$s bash scripts/prof_sad.sh | grep "candidates\|exact_rows"; done
