#!/bin/bash
exec bash "$(dirname "$0")/run.sh" test.py "$@"
