// bt2g_build_main.cpp -- bowtie2-build-s / bowtie2-build-l: the reference's index builder command line
// (bt2_build.cpp:105-330, bowtie_build_main.cpp) on top of the GPU builder in libbt2g.so.  The index width follows
// the executable's name, as the reference's wrapper script expects (bowtie2-build:66-80); --large-index forces .bt2l.
#include <cstring>
#include <string>

int bt2g_build_cli_main(int argc, const char** argv, int large_default);

int main(int argc, const char** argv) {
	const std::string me = argv[0];
	const size_t sl = me.find_last_of('/');
	const std::string base = sl == std::string::npos ? me : me.substr(sl + 1);
	const bool large = base.find("build-l") != std::string::npos;
	return bt2g_build_cli_main(argc, argv, large ? 1 : 0);
}
