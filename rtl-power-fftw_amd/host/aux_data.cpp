#include "aux_data.h"

#include <fstream>
#include <iostream>
#include <sstream>

namespace rpf_host {

template <typename T>
std::vector<T> read_value_column(std::istream& in)
{
    std::vector<T> column;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream fields(line);
        if ((fields >> std::ws).peek() == '#') continue;     // comment line
        T field, last{};
        bool any = false;
        while (fields >> field) {
            last = field;
            any = true;
        }
        if (any) column.push_back(last);                      // blank / text-only lines are skipped
    }
    return column;
}
template std::vector<float> read_value_column<float>(std::istream&);
template std::vector<double> read_value_column<double>(std::istream&);

namespace {

template <typename T>
std::vector<T> read_named(const std::string& name, const char* what, int expected, std::istream& standard_input)
{
    std::vector<T> values;
    if (name == "-") {
        std::cerr << "Reading " << what << " from stdin." << std::endl;
        values = read_value_column<T>(standard_input);
    } else {
        std::cerr << "Reading " << what << " from file " << name << std::endl;
        std::ifstream file(name);
        if (!file.good())
            throw RPFexception("Could not open " + name + ". Quitting.", ReturnValue::InvalidInput);
        values = read_value_column<T>(file);
    }
    if (static_cast<int>(values.size()) != expected) {
        const std::string label = std::string(what) == "baseline" ? "baseline" : "window function";
        throw RPFexception("Error reading " + label + ". Expected " + std::to_string(expected) +
                           " values, found " + std::to_string(values.size()) + ".",
                           ReturnValue::InvalidInput);
    }
    std::cerr << "Succesfully read " << values.size() << " " << what << " points." << std::endl;
    return values;
}

}  // namespace

AuxData::AuxData(const Options& options) { load(options, std::cin); }
AuxData::AuxData(const Options& options, std::istream& standard_input) { load(options, standard_input); }

void AuxData::load(const Options& o, std::istream& standard_input)
{
    if (o.window && o.baseline && o.window_file == "-" && o.baseline_file == "-") {
        // both on stdin: 2N values, baseline first, then the window (man page)
        std::cerr << "Reading baseline and window function from stdin." << std::endl;
        const std::vector<double> all = read_value_column<double>(standard_input);
        if (static_cast<int>(all.size()) != 2 * o.N)
            throw RPFexception("Error reading window function and baseline from stdin. Expected " +
                               std::to_string(2 * o.N) + " values, found " + std::to_string(all.size()) + ".",
                               ReturnValue::InvalidInput);
        baseline_values.assign(all.begin(), all.begin() + o.N);
        for (int i = 0; i < o.N; ++i) window_values.push_back(static_cast<float>(all[o.N + i]));
        std::cerr << "Succesfully read " << window_values.size() << " window function points." << std::endl;
        std::cerr << "Succesfully read " << baseline_values.size() << " baseline points." << std::endl;
        return;
    }
    if (o.window) window_values = read_named<float>(o.window_file, "window function", o.N, standard_input);
    if (o.baseline) baseline_values = read_named<double>(o.baseline_file, "baseline", o.N, standard_input);
}

}  // namespace rpf_host
