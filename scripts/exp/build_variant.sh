#!/bin/bash
# build_ab/lib_<name>.so: the library with ONE source file rebuilt under extra -D flags.  usage: build_variant.sh <name> <file.hip> <flags...>
set -e
cd "$(dirname "$0")/../../whisprrec_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../../build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -I../../include -Wno-unused-function "$@" -c $src -o /tmp/variant_$name.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/lib_$name.so $objs /tmp/variant_$name.o
