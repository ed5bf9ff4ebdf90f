#!/usr/bin/env python3
"""Golden vectors for the per-vertex angle excesses of the land_water surface mesh (CVX_MeshUtil::computeShapeComplexity,
LW/VX_MeshUtil.cpp:956-1031), produced by the UNMODIFIED reference binary oracle/_ref/voxelyze_lw_ref.  Build container only.

The reference writes the vector to <CurvaturesTmpFile>, runs `python <...>/curvatureEntropy.py <file>` (a script its repository does
not contain: the call fails harmlessly), sleeps a second, reads a number back and removes the file with `rm <file>`.  Two facts make
the file capturable without touching the reference:
  * a path with a SPACE in it is opened as one name by the C++ stream but reaches `rm` (and `python`) as two words, so the file
    survives: after the run it holds the vector of the FINAL state (computeFinalShapeComplexity, main.cpp:117);
  * the vector of the REST state (computeInitialShapeComplexity, main.cpp:65) is written before the first time step and sits there for
    the second the reference sleeps: it is copied as soon as it is complete (it ends with a tab after the last of `count` values).
Writes tests/golden/expected/<case>.curv_start.txt / .curv_end.txt."""
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_BIN = os.path.join(REPO, "oracle", "_ref", "voxelyze_lw_ref")
CASES = ["lw_swim6", "lw_land6", "lw_stiff5"]


def main():
    sys.path.insert(0, REPO)
    from evosoro_amd import engine
    for case in CASES:
        text = open(os.path.join(HERE, "vxa", case + ".vxa")).read()
        work = tempfile.mkdtemp(prefix="curv gold ")            # (the space that keeps `rm` from removing the file)
        os.makedirs(os.path.join(work, "fitnessFiles"))
        curv = os.path.join(work, "curv file.txt")
        text = re.sub(r"<CurvaturesTmpFile>.*?</CurvaturesTmpFile>", "<CurvaturesTmpFile>%s</CurvaturesTmpFile>" % curv, text)
        text = re.sub(r"<FitnessFileName>.*?</FitnessFileName>", "<FitnessFileName>%s</FitnessFileName>" % os.path.join(work, "fitnessFiles", "out.xml"), text)
        text = re.sub(r"<QhullTmpFile>.*?</QhullTmpFile>", "<QhullTmpFile>%s</QhullTmpFile>" % os.path.join(work, "qhull.txt"), text)
        vxa = os.path.join(work, "in.vxa")
        open(vxa, "w").write(text)
        count = len(engine.inspect_angle_excess(text))          # (number of mesh vertices: when the file is complete)
        proc = subprocess.Popen([REF_BIN, "-f", vxa], cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        start = None
        t0 = time.time()
        while start is None and time.time() - t0 < 180:      # (the qhull attempts before it take half a minute when qhull is not installed)
            try:
                data = open(curv).read()
                if data.endswith("\t") and len(data.split()) == count:
                    start = data
            except IOError:
                pass
            time.sleep(0.005)
        proc.wait(timeout=600)
        end = open(curv).read()
        assert start is not None and len(end.split()) == count, case
        assert os.path.exists(os.path.join(work, "fitnessFiles", "out.xml")), case
        open(os.path.join(HERE, "expected", case + ".curv_start.txt"), "w").write(start)
        open(os.path.join(HERE, "expected", case + ".curv_end.txt"), "w").write(end)
        print(case, count, "vertices; start[:4]", start.split()[:4], "end[:4]", end.split()[:4])
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
