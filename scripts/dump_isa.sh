#!/bin/bash
# Dumps the gfx950 ISA of the C2 encode/decode kernels to /tmp/asm/{enc,dec}.s   (extra args: -D flags)
set -e
mkdir -p /tmp/asm && cd /tmp/asm
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -save-temps "$@" -c /root/repo/constriction_amd/csrc/cst_api.hip -o /tmp/asm/cst_api.o 2>/dev/null
S=cst_api-hip-amdgcn-amd-amdhsa-gfx950.s
L=$(grep -n "^_ZN3cst17ans_encode_kernelILi32ELi64ELi0ELb1ELi8ELb1EEEvNS_13AnsEncodeArgsE:" $S | cut -d: -f1)
awk -v s=$L 'NR>=s && NR<=s+8000' $S | awk '/s_endpgm/{print; exit} {print}' > enc.s
L=$(grep -n "^_ZN3cst17ans_decode_kernelILi32ELi64ELi0ELb1ELi1ELb1ELi8ELb1EEEvNS_13AnsDecodeArgsE:" $S | cut -d: -f1)
awk -v s=$L 'NR>=s && NR<=s+8000' $S | awk '/s_endpgm/{print; exit} {print}' > dec.s
wc -l enc.s dec.s
