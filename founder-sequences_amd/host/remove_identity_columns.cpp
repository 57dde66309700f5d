// remove_identity_columns -- same command line surface and behaviour as the reference's tool
// (remove-identity-columns/cmdline.ggo:4-12, main.cc:160-227): read the aligned text paths from a list,
// write each text without the columns in which all texts agree to a file of the same base name in the
// current directory, and print the identity columns as a string of 0 / 1 (1 = identity) to stdout.
// Host only; nothing here touches the GPU path.
#include "aux_common.hpp"

#include <cstring>
#include <getopt.h>

int main(int argc, char **argv)
{
	char const *input = "-";
	bool overwrite = false;
	static option const opts[] = {{"input", required_argument, nullptr, 'i'}, {"overwrite", no_argument, nullptr, 1000},
	                              {"help", no_argument, nullptr, 'h'}, {"version", no_argument, nullptr, 'V'}, {nullptr, 0, nullptr, 0}};
	for (int c; (c = getopt_long(argc, argv, "i:hV", opts, nullptr)) != -1;)
		switch (c)
		{
			case 'i': input = optarg; break;
			case 1000: overwrite = true; break;
			case 'h':
				std::cout << "Usage: remove_identity_columns < input-list.txt > identity-columns.txt\n"
				             "  -i, --input=PATH   Input file path  (default=`-')\n      --overwrite    Overwrite the output files if needed  (default=off)\n";
				return EXIT_SUCCESS;
			case 'V': std::cout << "remove_identity_columns 0.1\n"; return EXIT_SUCCESS;
			default: return EXIT_FAILURE;
		}
	std::ios_base::sync_with_stdio(false);
	std::vector<std::string> in_names;
	if (0 == strcmp(input, "-")) aux::read_lines(std::cin, in_names);
	else
	{
		std::ifstream f(input);
		if (!f) { std::cerr << "Unable to open the input file '" << input << "'." << std::endl; return EXIT_FAILURE; }
		aux::read_lines(f, in_names);
	}
	std::vector<std::string> out_names;
	for (auto const &n : in_names) out_names.push_back(aux::base_name(n));

	std::cerr << "Creating the output files…" << std::endl;                     // main.cc:198-199
	for (auto const &n : out_names) if (!aux::create_output(n, overwrite)) return EXIT_FAILURE;

	std::cerr << "Handing input…" << std::endl;                                   // main.cc:201 (sic)
	// One 32 KiB chunk of every text at a time; a file is opened, read at its offset and closed again per chunk
	// (as the reference does, main.cc fill_buffers / output_buffer_contents), so that thousands of haplotype files
	// work under the default descriptor limit.
	size_t const chunk = 32 * 1024;                                             // main.cc:68
	std::vector<std::string> bufs(in_names.size());
	std::string mask, kept;
	for (std::streamoff offset = 0;; offset += (std::streamoff) chunk)
	{
		std::streamsize count = -1;
		for (size_t i = 0; i < in_names.size(); ++i)
		{
			std::ifstream in(in_names[i], std::ios::binary);
			if (!in) { std::cerr << "Unable to open '" << in_names[i] << "'." << std::endl; return EXIT_FAILURE; }
			in.seekg(offset);
			bufs[i].resize(chunk);
			in.read(&bufs[i][0], (std::streamsize) chunk);
			std::streamsize const got = in.gcount();
			if (-1 == count) count = got;
			else if (got != count) { std::cerr << "Got an unexpected number of characters from input." << std::endl; return EXIT_FAILURE; }   // main.cc:92-96
		}
		if (count <= 0) break;
		mask.assign((size_t) count, '0');
		size_t skipped = 0;
		for (std::streamsize k = 0; k < count; ++k)
		{
			bool same = true;
			for (size_t i = 1; i < bufs.size() && same; ++i) same = bufs[i][(size_t) k] == bufs[0][(size_t) k];
			if (same) { mask[(size_t) k] = '1'; ++skipped; }
		}
		if (skipped != (size_t) count)
			for (size_t i = 0; i < out_names.size(); ++i)
			{
				kept.clear();
				for (std::streamsize k = 0; k < count; ++k)
					if ('0' == mask[(size_t) k]) kept.push_back(bufs[i][(size_t) k]);
				std::ofstream out(out_names[i], std::ios::binary | std::ios::app);
				if (!out) { std::cerr << "Unable to open the output file '" << out_names[i] << "'." << std::endl; return EXIT_FAILURE; }
				out.write(kept.data(), (std::streamsize) kept.size());
				out.flush();
				if (!out) { std::cerr << "Unable to write to '" << out_names[i] << "'." << std::endl; return EXIT_FAILURE; }
			}
		std::cout << mask;
		if (count < (std::streamsize) chunk) break;
	}
	std::cout << std::endl;
	return EXIT_SUCCESS;
}
