"""Disassembly of the gfx950 code objects inside the built library (llvm-objdump of the ROCm toolchain): lets a CPU test
check properties of the SHIPPED instructions, not of the source."""
import os
import re
import shutil
import subprocess
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


def disassemble_library(lib_path):
    """-> {kernel symbol: [instruction lines]} over every gfx950 code object bundled in `lib_path`."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        lib = os.path.join(tmp, os.path.basename(lib_path))
        shutil.copy(lib_path, lib)            # --offloading writes the extracted bundles next to its input
        subprocess.run([OBJDUMP, '--offloading', lib], check=True, capture_output=True, cwd=tmp)
        for name in sorted(os.listdir(tmp)):
            if not name.endswith('gfx950'):
                continue
            text = subprocess.run([OBJDUMP, '-d', os.path.join(tmp, name)], check=True, capture_output=True,
                                  text=True).stdout
            sym = None
            for line in text.splitlines():
                m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
                if m:
                    sym = m.group(1)
                    out.setdefault(sym, [])
                elif sym and line.startswith('\t'):
                    out[sym].append(line.split('//')[0].strip())
    return out


def packed_fp32_with_op_sel(lib_path):
    """Packed-fp32 VOP3P instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose source selection is not the
    default - the operand form measured to go wrong beside bf16 MFMAs on MI355X."""
    bad = []
    for sym, lines in disassemble_library(lib_path).items():
        for ins in lines:
            if re.match(r'v_pk_(fma|mul|add)_f32\b', ins) and 'op_sel' in ins:
                bad.append((sym, ins))
    return bad
