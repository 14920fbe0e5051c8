// bt2g_main.cpp -- bowtie2-align-s / bowtie2-align-l: main() around the library's bowtie() (bt2g_search.cpp), the way the
// reference's bowtie_main.cpp:30-67 wraps bt2_search.cpp:5223.
extern "C" int bowtie(int argc, const char** argv);
int main(int argc, char** argv) { return bowtie(argc, const_cast<const char**>(argv)); }
