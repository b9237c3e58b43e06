"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU oracle for the StarDist prediction hot path:

* ``oracle/_ref``  : the reference's own native modules (stardist/lib/stardist2d.cpp,
  stardist3d.cpp, stardist3d_impl.cpp + vendored Clipper/Qhull/nanoflann), compiled
  from the sources where they lie under /root/reference by ``oracle/Makefile``.
  Loaded through :mod:`oracle.ref`.
* ``oracle/port.py``: numpy restatement of the reference's Python glue that cannot be
  imported here (stardist/nms.py, geometry/geom2d.py, geometry/geom3d.py, matching.py,
  scikit-image's polygon rule), each function citing the reference file:line.
* ``oracle/synth.py``: the seeded synthetic generators of SURVEY.md section 8(d).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import anything from this package.  The product (``stardist_amd``) never does.
"""
