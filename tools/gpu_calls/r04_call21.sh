#!/bin/bash
# r04: split mode on ragged batches / other geometries (new test parametrizations)
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "split" 2>&1 | tail -6
