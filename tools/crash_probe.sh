probe() { for i in 1 2 3 4 5; do timeout 400 python -X faulthandler bench.py "$@" > /tmp/o.log 2> /tmp/e.log; rc=$?; echo "args=[$*] run $i rc=$rc bytes=$(wc -c < /tmp/o.log) $(grep -m1 -A3 'most recent call first' /tmp/e.log | tail -2 | tr '\n' ' ' | cut -c1-160)"; done; }
probe --subs c2,c3mix,c3mce,c4,c4mce,k1,c5w1 --no-cpu-baseline
