// insert_identity_columns -- same command line surface and behaviour as the reference's tool
// (insert-identity-columns/cmdline.ggo:4-15, main.cc:118-319): given founder sequences (one text file with a
// sequence per line, or a list of files), the reference sequence and the identity-column string written by
// remove_identity_columns, write the founders with the identity columns put back (files "1".."N" for text
// input, the input base names for a list file) into the current directory.  Host only.
#include "aux_common.hpp"

#include <cstring>
#include <getopt.h>

int main(int argc, char **argv)
{
	char const *input = nullptr, *reference = nullptr, *identity = nullptr;
	bool overwrite = false, list_file = false;
	static option const opts[] = {{"input", required_argument, nullptr, 'i'}, {"input-format", required_argument, nullptr, 'f'},
	                              {"reference", required_argument, nullptr, 'r'}, {"identity-columns", required_argument, nullptr, 'd'},
	                              {"overwrite", no_argument, nullptr, 1000}, {"help", no_argument, nullptr, 'h'}, {nullptr, 0, nullptr, 0}};
	for (int c; (c = getopt_long(argc, argv, "i:f:r:d:h", opts, nullptr)) != -1;)
		switch (c)
		{
			case 'i': input = optarg; break;
			case 'r': reference = optarg; break;
			case 'd': identity = optarg; break;
			case 'f':
				if (0 == strcmp(optarg, "text")) list_file = false;
				else if (0 == strcmp(optarg, "list-file")) list_file = true;
				else { std::cerr << argv[0] << ": invalid argument, \"" << optarg << "\", for option `--input-format' (`-f')" << std::endl; return EXIT_FAILURE; }
				break;
			case 1000: overwrite = true; break;
			case 'h':
				std::cout << "Usage: insert_identity_columns -i input-list.txt -r reference-sequence.txt -d identity-columns.txt\n";
				return EXIT_SUCCESS;
			default: return EXIT_FAILURE;
		}
	if (!input) { std::cerr << argv[0] << ": '--input' ('-i') option required" << std::endl; return EXIT_FAILURE; }
	if (!reference) { std::cerr << argv[0] << ": '--reference' ('-r') option required" << std::endl; return EXIT_FAILURE; }
	if (!identity) { std::cerr << argv[0] << ": '--identity-columns' ('-d') option required" << std::endl; return EXIT_FAILURE; }

	std::cerr << "Opening the files…" << std::endl;
	std::string ref, idc;
	if (!aux::read_file(reference, ref)) { std::cerr << "Unable to open the reference '" << reference << "'." << std::endl; return EXIT_FAILURE; }
	if (!aux::read_file(identity, idc)) { std::cerr << "Unable to open the identity columns '" << identity << "'." << std::endl; return EXIT_FAILURE; }

	std::vector<std::string> founders, out_names;
	if (list_file)                                                              // main.cc:206-231
	{
		std::vector<std::string> names;
		if (0 == strcmp(input, "-")) aux::read_lines(std::cin, names);
		else
		{
			std::ifstream f(input);
			if (!f) { std::cerr << "Unable to open the input file '" << input << "'." << std::endl; return EXIT_FAILURE; }
			aux::read_lines(f, names);
		}
		for (auto const &n : names)
		{
			founders.emplace_back();
			if (!aux::read_file(n, founders.back())) { std::cerr << "Unable to open '" << n << "'." << std::endl; return EXIT_FAILURE; }
			out_names.push_back(aux::base_name(n));
		}
	}
	else                                                                        // main.cc:234-270
	{
		if (0 == strcmp(input, "-")) { std::cerr << "Memory mapping needed for single-file input." << std::endl; return EXIT_FAILURE; }
		std::string all;
		if (!aux::read_file(input, all)) { std::cerr << "Unable to open the input file '" << input << "'." << std::endl; return EXIT_FAILURE; }
		size_t const len = all.find_first_of('\n');
		if (std::string::npos == len) { std::cerr << "The input does not end a sequence with a newline." << std::endl; return EXIT_FAILURE; }
		for (size_t i = len; i < all.size(); i += 1 + len)
		{
			if (all[i] != '\n') { std::cerr << "The sequences in the input do not have the same length." << std::endl; return EXIT_FAILURE; }
			founders.push_back(all.substr(i - len, len));
			out_names.push_back(std::to_string(founders.size()));                 // files 1 .. N
		}
	}
	std::vector<std::ofstream> outs(founders.size());
	for (size_t i = 0; i < founders.size(); ++i)
	{
		if (!aux::create_output(out_names[i], overwrite)) return EXIT_FAILURE;
		outs[i].open(out_names[i], std::ios::binary | std::ios::trunc);
	}

	std::cerr << "Handling the input…" << std::endl;                             // main.cc:138-195
	size_t aligned = 0, taken = 0;
	for (char const is_identity : idc)
	{
		if ('\n' == is_identity) break;
		if ('0' == is_identity)
		{
			for (size_t i = 0; i < founders.size(); ++i)
			{
				if (taken >= founders[i].size()) { std::cerr << "A founder sequence is shorter than the identity columns require." << std::endl; return EXIT_FAILURE; }
				outs[i].put(founders[i][taken]);
			}
			++taken;
		}
		else if ('1' == is_identity)
		{
			if (aligned >= ref.size()) { std::cerr << "The reference is shorter than the identity columns require." << std::endl; return EXIT_FAILURE; }
			for (auto &o : outs) o.put(ref[aligned]);
		}
		else { std::cerr << "Unexpected character" << std::endl; return EXIT_FAILURE; }
		++aligned;
	}
	for (auto &o : outs) o.flush();
	return EXIT_SUCCESS;
}
