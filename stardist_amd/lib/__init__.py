"""Native layer of stardist_amd: mirrors the reference's `stardist.lib` package
(`stardist.lib.stardist2d`, `stardist.lib.stardist3d`) on top of libstardist_hip.so."""
