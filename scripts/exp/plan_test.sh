cd "$(dirname "$0")"
for v in 1; do
./xerr_exp 10000 20000 $v 30 3 | grep variant
./xerr_exp 8192 20000 $v 30 4 | grep variant
./xerr_exp 1808 20000 $v 30 17 | grep variant
./xerr_exp 1808 20000 $v 30 12 | grep variant
./xerr_exp 10000 20000 $v 30 3 | grep variant
./xerr_exp 20000 10000 $v 30 3 | grep variant
./xerr_exp 16384 10000 $v 30 2 | grep variant
./xerr_exp 3616 10000 $v 30 8 | grep variant
done
