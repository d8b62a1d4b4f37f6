fails=0
for seed in 410 411 412 413 414 415 416 417; do
  for w in 2 3; do
    port=$((29700 + seed % 50 + w))
    pids=""
    for r in $(seq 0 $((w-1))); do timeout -k 5 150 python tests/sharded_worker.py $r $w $port gloo $seed > gpurun_out/sw_${seed}_${w}_$r.log 2>&1 & pids="$pids $!"; done
    ok=1; for p in $pids; do wait $p || ok=0; done
    if [ $ok -eq 0 ]; then fails=$((fails+1)); echo "seed $seed world $w FAILED"; tail -3 gpurun_out/sw_${seed}_${w}_0.log; fi
  done
done
echo "sharded stress: $fails failures"
