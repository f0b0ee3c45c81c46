// collada_dump <scene.dae> -- prints the flat scene lf_collada.cpp builds, in the format of
// `ref_dump collada` (oracle/ref_driver.cpp), for tests/test_collada_loader.py.
#include <cstdio>

#include "lf_collada.h"

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: collada_dump <scene.dae>\n"); return 2; }
  lfamd::ColladaScene scene;
  std::string err;
  if (!lfamd::load_collada(argv[1], scene, err)) { std::fprintf(stderr, "collada_dump: %s\n", err.c_str()); return 1; }
  std::fputs(lfamd::dump_collada(scene).c_str(), stdout);
  return 0;
}
