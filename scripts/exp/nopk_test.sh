cd "$(dirname "$0")"
for r in 1 2 3 4; do
./xerr_exp 20000 10000 50 60 | grep "^variant"
./xerr_exp_nopk 20000 10000 50 60 | grep "^variant"
done
