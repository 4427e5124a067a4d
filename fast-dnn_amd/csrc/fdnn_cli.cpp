// fdnn_cli.cpp -- `fast-dnn model.bin input.bin [out] [BIN|TXT]` on one MI355X.
//
// Same command line, console lines and output files as the reference CLI
// (src/cpp/dnn.cc:20-84; BatchData file reader float_dnn.cc:85-105, dump /
// dumpToFile float_dnn.cc:114-164): input matrix big-endian `i32 n, i32 dim,
// n*dim f32`; BIN output host-endian `u32 n, u32 dim` + raw floats; TXT output
// one frame per line written with ostream's default float format; no out-path
// prints "%f" values to stdout.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/fdnn.h"

namespace {

bool read_feature_matrix(const std::string &path, std::vector<float> *data, int *n, int *dim) {
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) return false;
  unsigned char hdr[8];
  if (std::fread(hdr, 1, 8, fp) != 8) {
    std::fclose(fp);
    return false;
  }
  auto be = [](const unsigned char *p) {
    return int32_t((uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]));
  };
  *n = be(hdr);
  *dim = be(hdr + 4);
  if (*n < 0 || *dim <= 0) {
    std::fclose(fp);
    return false;
  }
  const size_t count = size_t(*n) * size_t(*dim);
  std::vector<unsigned char> raw(count * 4);
  const bool ok = std::fread(raw.data(), 1, raw.size(), fp) == raw.size();
  std::fclose(fp);
  if (!ok) return false;
  data->resize(count);
  for (size_t i = 0; i < count; ++i) {
    uint32_t v = uint32_t(be(&raw[i * 4]));
    std::memcpy(&(*data)[i], &v, 4);
  }
  return true;
}

}  // namespace

int main(int argc, char *argv[]) {
  using std::cout;
  using std::endl;
  if (argc < 3) {
    cout << "At least two parameters are required. "
            "[model-path] [binary-input-path] Optional[out-path] Optional[out-type BIN|TXT]"
         << endl;
    return -1;
  }
  const std::string model_path = argv[1], input_path = argv[2];
  const std::string output_path = argc > 3 ? argv[3] : "";
  const std::string out_type = argc > 4 ? argv[4] : "";
  cout << "Model File  = " << model_path << endl;
  cout << "Input File  = " << input_path << endl;
  if (!output_path.empty()) cout << "Output File = " << output_path << endl;
  if (!out_type.empty()) cout << "Output Type = " << out_type << endl;
  bool binary = false;
  if (!out_type.empty()) {
    binary = out_type == "BIN";
    if (!binary && out_type != "TXT") {
      cout << "Unidentified output file type = " << out_type;
      return -1;
    }
  }

  fdnn_model *model = nullptr;
  if (fdnn_model_load(model_path.c_str(), 3.0f, &model) != FDNN_OK) {
    std::cerr << "fast-dnn: " << fdnn_last_error() << endl;
    return 3;
  }
  const int O = fdnn_model_output_dim(model);
  // PrintTopology (float_dnn.cc:71-74) prints layers-2 as the hidden count
  cout << "Network = " << fdnn_model_input_dim(model) << "-" << fdnn_model_layer_count(model) - 2 << "x"
       << fdnn_model_hidden_dim(model) << "-" << O << endl;

  std::vector<float> input;
  int n = 0, dim = 0;
  if (!read_feature_matrix(input_path, &input, &n, &dim)) {
    std::cerr << "fast-dnn: cannot read input matrix " << input_path << endl;
    return 3;
  }
  cout << "Input   = " << n << "x" << dim << endl;

  std::vector<float> out(size_t(n) * size_t(O));
  const auto t0 = std::chrono::high_resolution_clock::now();
  if (fdnn_calculate(model, input.data(), n, dim, 8, out.data()) != FDNN_OK) {
    std::cerr << "fast-dnn: " << fdnn_last_error() << endl;
    return 3;
  }
  const auto t1 = std::chrono::high_resolution_clock::now();
  cout << "Dnn calculation time = " << std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count() << " ms."
       << endl;

  if (output_path.empty()) {
    const float *p = out.data();
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < O; ++j) {
        std::printf("%f", *p++);
        if (j < O - 1) cout << " ";
      }
      cout << endl;
    }
  } else {
    std::ofstream os;
    if (binary)
      os.open(output_path, std::ios::binary | std::ios::out);
    else
      os.open(output_path);
    if (!os.is_open()) {
      cout << "Cannot open file " << output_path << endl;
    } else {
      if (binary) {
        const uint32_t v = uint32_t(n), d = uint32_t(O);
        os.write(reinterpret_cast<const char *>(&v), 4);
        os.write(reinterpret_cast<const char *>(&d), 4);
        os.write(reinterpret_cast<const char *>(out.data()), std::streamsize(out.size() * sizeof(float)));
      } else {
        const float *p = out.data();
        for (int i = 0; i < n; ++i) {
          for (int j = 0; j < O; ++j) {
            os << *p++;
            if (j < O - 1) os << " ";
          }
          os << endl;
        }
      }
    }
  }
  fdnn_model_free(model);
  return 0;
}
