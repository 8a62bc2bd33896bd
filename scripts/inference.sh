#!/bin/bash
exec bash "$(dirname "$0")/run.sh" inference.py "$@"
