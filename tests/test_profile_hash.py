"""the hash that ties a profile's counter traffic to a kernel's sources (cortex.jl_amd/build.py: sources_sha16) looks at code only:
comments and blank lines may change under a committed profile, code may not"""
from importlib import import_module

build = import_module("cortex.jl_amd.build")


def test_comments_and_blank_lines_do_not_count():
    a = 'int a = 1;   // one\n\n/* a block\n   comment */ int b = 2;\nconst char *u = "http://x"; // trailing\n'
    b = 'int a = 1;\n int b = 2;\nconst char *u = "http://x";\n'
    assert build._code_only(a).split() == build._code_only(b).split()
    assert build._code_only(a) != build._code_only(a.replace("b = 2", "b = 3"))


def test_every_profiled_kernel_family_hashes():
    for k in ("k_sweep", "k_chain_onepass", "k_rule64w", "k_mvc_apply", "k_mf_normal", "k_pscan_totals"):
        h = build.sources_sha16(k)
        assert len(h) == 16 and int(h, 16) >= 0
