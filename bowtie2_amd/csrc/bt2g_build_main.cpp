// bt2g_build_main.cpp -- bowtie2-build-s / bowtie2-build-l: main() around the library's bowtie_build() (bt2g_build.hip), the way the
// reference's bowtie_build_main.cpp wraps bt2_build.cpp:560.  The index width follows the executable's name, as the reference's
// wrapper script expects (bowtie2-build:66-80); --large-index forces .bt2l.
extern "C" int bowtie_build(int argc, const char** argv);
int main(int argc, const char** argv) { return bowtie_build(argc, argv); }
