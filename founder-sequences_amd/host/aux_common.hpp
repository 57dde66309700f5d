// aux_common.hpp -- helpers shared by the three auxiliary command line tools.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

namespace aux {

// The tools name their outputs after the last path component of the inputs (regex ".*?([^/]+)$" in
// remove-identity-columns/main.cc:186 and insert-identity-columns/main.cc:216).
inline std::string base_name(std::string const &path)
{
	size_t const p = path.find_last_of('/');
	return p == std::string::npos ? path : path.substr(p + 1);
}

inline bool read_lines(std::istream &is, std::vector<std::string> &out)
{
	std::string line;
	while (std::getline(is, line)) out.push_back(line);
	return true;
}

inline bool read_file(std::string const &path, std::string &out)
{
	std::ifstream f(path, std::ios::binary);
	if (!f) return false;
	out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
	return true;
}

// lb::open_file_for_writing with CREATE and optionally OVERWRITE: without --overwrite an existing file is an error
inline bool create_output(std::string const &path, bool overwrite)
{
	if (!overwrite)
	{
		std::ifstream probe(path);
		if (probe) { std::cerr << "Unable to create the output file '" << path << "': it exists (use --overwrite)." << std::endl; return false; }
	}
	std::ofstream f(path, std::ios::binary | std::ios::trunc);
	if (!f) { std::cerr << "Unable to create the output file '" << path << "'." << std::endl; return false; }
	return true;
}

} // namespace aux
