#!/bin/bash
# Build a diagnostic variant quickly: copy the default objects, recompile only the named sources with extra flags.
#   bash scripts/mkvar.sh NAME "FLAGS" file1.hip [file2.hip ...]   ->  m2_mixer_amd/libm2mixer_exp_NAME.so
set -e
cd "$(dirname "$0")/../m2_mixer_amd/csrc"
name=$1; flags=$2; shift 2
rm -rf build_exp_$name; mkdir -p build_exp_$name
cp -p build/*.o build_exp_$name/
for f in "$@"; do rm -f build_exp_$name/${f%.hip}.o; done
make -s EXP=$name EXPFLAGS="$flags" > /tmp/mk_$name.log 2>&1 || { tail -20 /tmp/mk_$name.log; exit 1; }
echo "built $name"
