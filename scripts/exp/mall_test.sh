cd "$(dirname "$0")"
# two passes per timed launch pair: 61 = both ascending (control), 60 = ascending then descending; sizes around the 256 MB infinity cache
for nm in "20000 10000" "8192 8192" "4096 8192" "8192 4096" "16384 8192"; do
for v in 61 60 61 60; do ./xerr_exp $nm $v 30 3 | grep "^variant"; done
done
