#!/bin/bash
# pytest with a -k expression given as ONE argument where '+' stands for ' or ' (tools/gpu_session.sh splits its step strings on blanks)
# Usage: tools/pytest_k.sh "<expr with + for or>" <pytest args...>
k=${1//+/ or }; shift
exec python -m pytest -q -m gpu -k "$k" "$@"
