cd "$(dirname "$0")"
for r in 1 2 3; do
./xerr_exp 20000 10000 50 40 | grep "^variant"
./xerr_exp_prio1 20000 10000 50 40 | grep "^variant"
./xerr_exp_prio3 20000 10000 50 40 | grep "^variant"
done
