"""The drop-in claim under test: the reference's OWN `main()` — the text of
src/common/bicg.cpp from `int main` to the end of the file, read from the reference checkout at test time, never stored
here — compiles and links as written behind `#include "mgs_host.hpp"` + `using namespace mgs;`.  Build container only
(skipped where /root/reference is absent, i.e. on the GPU box); nothing is executed: no GPU here."""
import os
import re
import subprocess

import pytest

from conftest import REPO

REF = "/root/reference/src/common/bicg.cpp"


@pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not present (GPU box)")
def test_reference_main_compiles_and_links_against_mgs_host(tmp_path):
    text = open(REF).read()
    m = re.search(r"^int main\s*\(", text, flags=re.M)
    assert m, "reference main() not found"
    main_src = text[m.start():]
    assert "BiCGSTABiml(A, x, b, precond, max_iter, tol)" in main_src and "readMatrix(" in main_src and "VectorXd" in main_src
    tu = tmp_path / "ref_main_tu.cpp"
    tu.write_text('#include <cstdio>\n#include <cstdlib>\n#include <iostream>\n#include <string>\n'
                  f'#include "{os.path.join(REPO, "multigridsolver_amd", "cpp", "mgs_host.hpp")}"\n'
                  "using namespace std;\nusing namespace mgs;\n" + main_src)
    lib = os.path.join(REPO, "multigridsolver_amd")
    assert os.path.exists(os.path.join(lib, "libmgs.so")), "libmgs.so not built (run __graft_entry__.build())"
    exe = tmp_path / "ref_main"
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-o", str(exe), str(tu), f"-L{lib}", "-lmgs", f"-Wl,-rpath,{lib}"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    assert exe.exists()
    # usage path (argc != 3) runs without a device: same message and exit code as the reference's main
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Incorrect number of arguments." in r.stdout
