"""CPU: the development knobs of libldm_hip.so are declared in ONE table (csrc/ldm_knobs.h) — every name the sources read
through knob_env / knob_int is in it (ldm_create refuses a stray one only if it is listed), every listed name is read
somewhere (no dead documentation: ADVICE r4), and INTEGRATION.md section 5 lists them all."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "layout_dm_amd", "csrc")


def _table():
    src = open(os.path.join(CSRC, "ldm_knobs.h")).read()
    body = src[src.index("knob_table()"):src.index("return t;")]
    return set(re.findall(r'\{"(LDM_[A-Z0-9_]+)"', body))


def _used():
    used = set()
    for f in os.listdir(CSRC):
        if f == "ldm_knobs.h":
            continue
        src = open(os.path.join(CSRC, f)).read()
        used |= set(re.findall(r'knob_(?:env|int)\(\s*"(LDM_[A-Z0-9_]+)"', src))
    return used


def test_every_knob_read_is_declared_and_every_declared_knob_is_read():
    table, used = _table(), _used()
    assert used - table == set(), f"read but not in knob_table (ldm_create would not refuse them): {sorted(used - table)}"
    assert table - used == set(), f"in knob_table but read nowhere: {sorted(table - used)}"
    # no source reads a LDM_* variable behind the table's back
    for f in os.listdir(CSRC):
        src = open(os.path.join(CSRC, f)).read()
        for name in re.findall(r'getenv\(\s*"(LDM_[A-Z0-9_]+)"', src):
            assert name == "LDM_DEV", (f, name)


def test_integration_doc_lists_every_knob():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [k for k in sorted(_table()) if f"`{k}" not in doc]
    assert not missing, f"INTEGRATION.md section 5 does not mention: {missing}"
