#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for i in 1 2 3 4 5 6; do timeout 3000 python3 -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -1; done
