for L in 64 32 16 8; do echo "== KLHR_FIT_LANES=$L"; KLHR_FIT_LANES=$L timeout -k 10 100 python tools/dbg_ktimes.py ill-normal 100 16384 || exit 1; done
for L in 64 32 16; do echo "== KLHR_FIT_LANES=$L"; KLHR_FIT_LANES=$L timeout -k 10 100 python tools/dbg_ktimes.py funnel 100 131072 || exit 1; done
