#!/bin/bash
# r04: relation tests after the packed-range guard (+ the new general-form test)
timeout 900 python -m pytest tests -m gpu -q -x -k "relation or config5" 2>&1 | tail -3
