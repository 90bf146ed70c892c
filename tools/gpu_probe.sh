#!/bin/bash
export TMPDIR=/tmp
timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
