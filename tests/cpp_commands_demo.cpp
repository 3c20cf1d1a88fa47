// Drives vectordb_amd/cpp/vdb_commands.hpp the way the reference's CommandProcessor does
// (command_processor.hpp:29-50, :237-248): a table of lower-case command names -> handlers taking a CommandContext
// (args, response bytes), dispatched by process_command; replies printed as text for the pytest.
//   -DVDB_REFERENCE_SERIALIZER -I/root/reference: the handlers answer through the reference's OWN ResponseSerializer
//     (response_serializer.hpp) and the bytes are decoded here by its wire format; build container only, the
//     reference does not travel to the GPU box.
//   otherwise: a serializer with the same static interface that writes that text directly.
// Both builds must print the same text for the same commands (tests/test_cpp_commands.py).
//   argv[1] = "host": parsing, errors, VADD / VCARD / VDEL (no GPU);  "gpu": VSEARCH as well;  argv[2]: where to save the index.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <random>
#include <string_view>
#include <unordered_map>

#ifdef VDB_REFERENCE_SERIALIZER
#include <iostream>  // response_serializer.hpp:81 writes to std::cout and includes no <iostream>
#include "response_serializer.hpp"
using Serializer = ResponseSerializer;
// the reference's wire format, response_serializer.hpp:31-58 / common.hpp:26-32 -> the text of TextSerializer below
static std::string decode(const std::vector<uint8_t>& b) {
    std::string out;
    size_t i = 0;
    while (i < b.size()) {
        const auto tag = static_cast<SerializationType>(b[i++]);
        if (tag == SerializationType::Nil) {
            out += "nil\n";
        } else if (tag == SerializationType::Error) {  // i32 code, u32 size, bytes
            int32_t code;
            uint32_t size;
            std::memcpy(&code, b.data() + i, 4);
            std::memcpy(&size, b.data() + i + 4, 4);
            // (its own deserialize_error reads the same layout)
            if (i == 1 && ResponseSerializer::deserialize_error(b) != std::string(b.begin() + 9, b.begin() + 9 + size)) std::abort();
            out += "err " + std::to_string(code) + " " + std::string(b.begin() + i + 8, b.begin() + i + 8 + size);
            i += 8 + size;
        } else if (tag == SerializationType::String) {
            uint32_t size;
            std::memcpy(&size, b.data() + i, 4);
            out += "str " + std::string(b.begin() + i + 4, b.begin() + i + 4 + size) + "\n";
            i += 4 + size;
        } else if (tag == SerializationType::Integer) {  // decimal text up to "\r\n" (:46-50)
            size_t e = i;
            while (e + 1 < b.size() && !(b[e] == '\r' && b[e + 1] == '\n')) ++e;
            out += "int " + std::string(b.begin() + i, b.begin() + e) + "\n";
            i = e + 2;
        } else {  // Double: 8 raw bytes
            double v;
            std::memcpy(&v, b.data() + i, 8);
            char buf[64];
            std::snprintf(buf, sizeof buf, "dbl %a\n", v);
            out += buf;
            i += 8;
        }
    }
    return out;
}
constexpr int ERR_ARG = -1, ERR_UNKNOWN = -2;  // command_processor.hpp:20-22
#else
constexpr int ERR_ARG = -1, ERR_UNKNOWN = -2;
#endif

#include "../vectordb_amd/cpp/vdb_commands.hpp"

#ifndef VDB_REFERENCE_SERIALIZER
struct TextSerializer {
    static void put(std::vector<uint8_t>& b, const std::string& s) { b.insert(b.end(), s.begin(), s.end()); }
    static void serialize_nil(std::vector<uint8_t>& b) { put(b, "nil\n"); }
    static void serialize_error(std::vector<uint8_t>& b, int32_t code, std::string_view msg) {
        put(b, "err " + std::to_string(code) + " " + std::string(msg));
    }
    static void serialize_string(std::vector<uint8_t>& b, std::string_view s) { put(b, "str " + std::string(s) + "\n"); }
    static void serialize_integer(std::vector<uint8_t>& b, int64_t v) { put(b, "int " + std::to_string(v) + "\n"); }
    static void serialize_double(std::vector<uint8_t>& b, double v) {
        char buf[64];
        std::snprintf(buf, sizeof buf, "dbl %a\n", v);
        put(b, buf);
    }
};
using Serializer = TextSerializer;
static std::string decode(const std::vector<uint8_t>& b) { return std::string(b.begin(), b.end()); }
#endif

// CommandProcessor::CommandContext without the KV store's EntryManager (command_processor.hpp:29-33)
struct CommandContext {
    const std::vector<std::string>& args;
    std::vector<uint8_t>& response;
};
static vdb_amd::VectorCommands<Serializer> vectors;  // beside command_handlers
static const std::unordered_map<std::string, std::function<void(CommandContext)>> command_handlers = {  // :237-248
    {"vadd", [](CommandContext c) { vectors.vadd(c.args, c.response); }},
    {"vsearch", [](CommandContext c) { vectors.vsearch(c.args, c.response); }},
    {"vdel", [](CommandContext c) { vectors.vdel(c.args, c.response); }},
    {"vcard", [](CommandContext c) { vectors.vcard(c.args, c.response); }},
};
static void process_command(CommandContext ctx) {  // :36-50
    if (ctx.args.empty()) return Serializer::serialize_error(ctx.response, ERR_ARG, "empty command\n");
    std::string command_key = ctx.args[0];
    std::transform(command_key.begin(), command_key.end(), command_key.begin(), ::tolower);
    auto it = command_handlers.find(command_key);
    if (it == command_handlers.end()) return Serializer::serialize_error(ctx.response, ERR_UNKNOWN, "unknown command\n");
    it->second(ctx);
}

int main(int argc, char** argv) {
    const bool gpu = argc > 1 && !std::strcmp(argv[1], "gpu");
    vectors.initial_capacity = 64;  // grows through resizeIndex
    auto run = [&](std::vector<std::string> args) {
        std::vector<uint8_t> out;
        process_command(CommandContext{args, out});
        std::string cmd = args.empty() ? "" : args[0];
        std::transform(cmd.begin(), cmd.end(), cmd.begin(), ::tolower);
        std::printf("> %s\n%s", cmd.c_str(), decode(out).c_str());
    };
    const size_t n = 300, dim = 8;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    std::vector<std::vector<std::string>> rows;
    for (size_t i = 0; i < n; ++i) {
        std::vector<std::string> a = {"vadd", "docs", std::to_string(1000 + i)};
        for (size_t c = 0; c < dim; ++c) {
            char buf[32];
            std::snprintf(buf, sizeof buf, "%.9g", u(rng));
            a.push_back(buf);
        }
        rows.push_back(a);
    }
    for (auto& a : rows) run(a);
    run({"vcard", "docs"});
    run({"vcard", "nothing"});
    run({"vadd", "docs", "5", "0.5"});                 // wrong dimension
    run({"vadd", "docs", "x", "1", "2"});              // bad label
    run({"vadd", "docs", "7", "1", "2", "zebra", "4", "5", "6", "7", "8"});
    run({"vdel", "docs", "1003"});
    run({"vdel", "docs", "999999"});
    run({"vsearch", "nothing", "3", "1", "2"});
    run({"VCARD", "docs"});   // names are matched in lower case (:41-42)
    run({"vfrobnicate", "docs"});
    run({});
    if (argc > 2) vectors.index("docs")->saveIndex(argv[2]);
    if (gpu) {
        for (size_t q = 0; q < 5; ++q) {
            std::vector<std::string> a = {"vsearch", "docs", "4"};
            a.insert(a.end(), rows[q].begin() + 3, rows[q].end());
            a.push_back("EF");
            a.push_back("50");
            run(a);
        }
    }
    return 0;
}
