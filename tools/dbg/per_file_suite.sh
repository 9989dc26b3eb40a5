mkdir -p gpurun_out/r06_perfile
for f in $(grep -l "mark.gpu" tests/*.py); do
  timeout 900 python -m pytest $f -m gpu -q -x > gpurun_out/r06_perfile/$(basename $f .py).log 2>&1
  echo "$f rc=$? $(tail -1 gpurun_out/r06_perfile/$(basename $f .py).log | cut -c1-100)"
done
