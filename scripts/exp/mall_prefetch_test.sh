cd "$(dirname "$0")"
for r in 1 2; do for v in 70 71 72 73; do ./xerr_exp 20000 10000 $v 30 3 | grep "^variant"; done; done
