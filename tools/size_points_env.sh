#!/bin/bash
# usage (GPU box): bash tools/size_points_env.sh "<ENV=VAL ...>" <model> <dim> <chains...>   (tools/size_points.sh under extra environment)
env_s=$1; shift
env $env_s bash $GRAFT_REPO_ROOT/tools/size_points.sh "$@"
