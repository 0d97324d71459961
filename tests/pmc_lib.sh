#!/bin/bash
# usage: pmc2.sh TAG LIBNAME "CTRS"
tag=$1; lib=$2; shift; shift
p=""; [ "$lib" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$lib.so
export SF_LIB_PATH=$p
$GRAFT_REPO_ROOT/tests/pmc.sh $tag "$@"
