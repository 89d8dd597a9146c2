cd "$GRAFT_REPO_ROOT"
for k in ${KS:-0.1 0.25 0.5 1 2}; do
  rm -f gpurun_out/parity_sample_report.json
  PORESEG_NOISE_K=$k python -m pytest "tests/test_parity_sample.py::test_sample_against_the_compiled_reference[filtered]" "tests/test_parity_sample.py::test_sample_against_the_compiled_reference[offgrid]" -q -m gpu 2>&1 | tail -1
  python - <<PY
import json
r=json.load(open('gpurun_out/parity_sample_report.json'))
for route,rep in r.items():
    d={c['name'] for c in rep['differing_cases']}
    print("K=$k", route, 'differing', len(d), 'warned', len(rep['near_tie_cases']), 'differing&warned', len(d & set(rep['near_tie_cases'])))
PY
done
