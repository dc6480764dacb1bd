"""Test / benchmark harness: stands in for the Rust caller of the render path.

  raytracer.hpp/.cpp   C++ twins of the reference's public scene types and util.rs helpers, each
                       with a describe() into an RtSceneDesc (what a Rust host adds to its traits)
  scenes.cpp           the reference's test scenes, transcribed; C entry points rth_*
  scenes.py            ctypes binding of libraytracer_host.so

Nothing here is on the render path: the product is rust-raytracer_amd/csrc + include/rt_amd.h."""
