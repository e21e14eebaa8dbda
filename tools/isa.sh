#!/bin/bash
# usage: tools/isa.sh <file.hip> <substring of the mangled kernel name>  -> /tmp/isa.s (that kernel's ISA) + its resource usage
cd /root/repo/multishiftseg_amd/csrc || exit 1
rm -rf /tmp/isa && mkdir -p /tmp/isa
timeout 600 hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -save-temps=obj -Rpass-analysis=kernel-resource-usage -c $1 -o /tmp/isa/out.o > /tmp/isa/log.txt 2>&1 < /dev/null
grep -A8 "Function Name.*$2" /tmp/isa/log.txt | grep "Name\|VGPRs:\|SGPRs:\|Scratch\|Occupancy\|LDS"
grep -i "error" /tmp/isa/log.txt | head
S=$(ls /tmp/isa/*gfx950.s 2>/dev/null | head -1)
[ -z "$S" ] && { echo "no .s produced"; ls /tmp/isa; exit 1; }
L=$(grep -n "^[_A-Za-z0-9]*$2[_A-Za-z0-9]*:" "$S" | head -1 | cut -d: -f1)
[ -z "$L" ] && { echo "kernel not found"; exit 1; }
tail -n +$L "$S" | awk '{print} /s_endpgm/{exit}' > /tmp/isa.s; wc -l /tmp/isa.s
