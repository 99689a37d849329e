#!/bin/bash
# Builds a variant of libepi.so with extra compiler flags into build/libepi_<name>.so (load it with EPI_LIB=...).
# usage: tools/build_variant.sh <name> [flags...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/epidemicsimulator_amd" "$root/build"
cp -r "$root/include" "$tmp/include"
cp -r "$root/epidemicsimulator_amd/csrc" "$tmp/epidemicsimulator_amd/csrc"
rm -f "$tmp"/epidemicsimulator_amd/csrc/*.o
make -s -C "$tmp/epidemicsimulator_amd/csrc" EXTRA="$*" >/dev/null
cp "$tmp/epidemicsimulator_amd/libepi.so" "$root/build/libepi_$name.so"
rm -rf "$tmp"
echo "build/libepi_$name.so"
