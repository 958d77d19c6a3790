// kitti_eval.cpp -- KITTI 2D-detection AP and Average Orientation Similarity
// (AOS), the IMAGE-metric path of the reference's offline evaluator, without
// Boost.  Host code.
//
// Reference: tools/kitti-eval/evaluate_object_3d_offline.cpp
//   loadDetections :131-176, loadGroundtruth :178-202, imageBoxOverlap :227-265,
//   getThresholds :346-379, cleanData :381-454, computeStatistics :456-615,
//   eval_class :622-706, 11-point summary in saveAndPlotPlots :720-724,
//   eval :791-850 (the IMAGE block; the ground / 3D blocks need Boost.Geometry
//   and are out of scope).
// The reference file cannot be compiled in this image (Boost headers absent), so
// this restatement is checked against an independent Python restatement
// (oracle/kitti_eval_oracle.py) and hand-computed cases: PARITY UNPINNED against
// the reference binary.
//
// Semantics kept on purpose: the class-overlap table as overwritten in the
// reference (0.7 / 0.5 / 0.5 for car / pedestrian / cyclist, :54), detection
// heights truncated to int (:441), "Van" / "Person_sitting" neighbours ignored,
// DontCare areas absorb unassigned detections (criterion = detection area),
// recall sampled at 41 points with the left/right-closest rule, precision and
// AOS made monotone from the right.
#include <dirent.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include <algorithm>
#include <string>
#include <vector>

namespace {

constexpr int kClasses = 3, kLevels = 3, kSamples = 41;
const char* const kClassName[kClasses] = {"car", "pedestrian", "cyclist"};
const int kMinHeight[kLevels] = {40, 25, 25};
const int kMaxOcclusion[kLevels] = {0, 1, 2};
const double kMaxTruncation[kLevels] = {0.15, 0.3, 0.5};
const double kMinOverlap[kClasses] = {0.7, 0.5, 0.5};  // MIN_OVERLAP[IMAGE][class]

struct Box {
  std::string type;
  double x1, y1, x2, y2, alpha;
};
struct Truth {
  Box box;
  double truncation;
  int occlusion;
};
struct Det {
  Box box;
  double score;
};
struct Frame {
  std::vector<Truth> gt;
  std::vector<Det> det;
};
struct Counts {
  std::vector<double> tp_scores;
  double similarity = 0;
  int tp = 0, fp = 0, fn = 0;
};

bool same(const std::string& a, const char* b) { return strcasecmp(a.c_str(), b) == 0; }

// mode -1: intersection over union; 0: over the first box's area
double overlap(const Box& a, const Box& b, int mode) {
  const double w = std::min(a.x2, b.x2) - std::max(a.x1, b.x1);
  const double h = std::min(a.y2, b.y2) - std::max(a.y1, b.y1);
  if (w <= 0 || h <= 0) return 0;
  const double inter = w * h;
  const double area_a = (a.x2 - a.x1) * (a.y2 - a.y1), area_b = (b.x2 - b.x1) * (b.y2 - b.y1);
  return mode == -1 ? inter / (area_a + area_b - inter) : inter / area_a;
}

bool read_lines(const std::string& path, std::vector<std::vector<std::string>>& rows) {
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return false;
  char buf[1024];
  while (fgets(buf, sizeof buf, f)) {
    std::vector<std::string> tok;
    for (char* p = strtok(buf, " \t\r\n"); p; p = strtok(nullptr, " \t\r\n")) tok.push_back(p);
    if (!tok.empty()) rows.push_back(tok);
  }
  fclose(f);
  return true;
}

// gt flags: 0 counted, 1 ignored (neighbour class / too hard), -1 other class
// det flags: 0 evaluated, 1 too small, -1 other class
void classify(int cls, int level, const Frame& fr, std::vector<int>& gt_flag, std::vector<int>& det_flag,
              std::vector<Box>& dontcare, int& n_gt) {
  for (const Truth& g : fr.gt) {
    int valid = -1;
    if (same(g.box.type, kClassName[cls])) valid = 1;
    else if (cls == 1 && same(g.box.type, "Person_sitting")) valid = 0;
    else if (cls == 0 && same(g.box.type, "Van")) valid = 0;
    const bool hard = g.occlusion > kMaxOcclusion[level] || g.truncation > kMaxTruncation[level] ||
                      (g.box.y2 - g.box.y1) < kMinHeight[level];
    if (valid == 1 && !hard) {
      gt_flag.push_back(0);
      ++n_gt;
    } else if (valid == 0 || (hard && valid == 1)) {
      gt_flag.push_back(1);
    } else {
      gt_flag.push_back(-1);
    }
    if (same(g.box.type, "DontCare")) dontcare.push_back(g.box);
  }
  for (const Det& d : fr.det) {
    const int height = (int)fabs(d.box.y1 - d.box.y2);  // truncated to an integer, as the reference does
    if (height < kMinHeight[level]) det_flag.push_back(1);
    else det_flag.push_back(same(d.box.type, kClassName[cls]) ? 0 : -1);
  }
}

Counts match(int cls, const Frame& fr, const std::vector<Box>& dontcare, const std::vector<int>& gt_flag,
             const std::vector<int>& det_flag, bool with_fp, bool with_aos, double thresh) {
  Counts out;
  const double kNone = -10000000;
  const size_t nd = fr.det.size();
  std::vector<char> taken(nd, 0), below(nd, 0);
  std::vector<double> delta;
  if (with_fp)
    for (size_t j = 0; j < nd; ++j) below[j] = fr.det[j].score < thresh;
  for (size_t i = 0; i < fr.gt.size(); ++i) {
    if (gt_flag[i] == -1) continue;
    int pick = -1;
    double valid = kNone, best = 0;
    bool picked_small = false;
    for (size_t j = 0; j < nd; ++j) {
      if (det_flag[j] == -1 || taken[j] || below[j]) continue;
      const double o = overlap(fr.det[j].box, fr.gt[i].box, -1);
      if (!(o > kMinOverlap[cls])) continue;
      if (!with_fp) {                       // recall pass: the most confident candidate
        if (fr.det[j].score > valid) {
          pick = (int)j;
          valid = fr.det[j].score;
        }
      } else if ((o > best || picked_small) && det_flag[j] == 0) {  // pr pass: the best-overlapping one
        best = o;
        pick = (int)j;
        valid = 1;
        picked_small = false;
      } else if (valid == kNone && det_flag[j] == 1) {
        pick = (int)j;
        valid = 1;
        picked_small = true;
      }
    }
    if (valid == kNone) {
      if (gt_flag[i] == 0) ++out.fn;
    } else if (gt_flag[i] == 1 || det_flag[pick] == 1) {
      taken[pick] = 1;                      // matched, but one side is ignored: neither TP nor FP
    } else {
      ++out.tp;
      out.tp_scores.push_back(fr.det[pick].score);
      if (with_aos) delta.push_back(fr.gt[i].box.alpha - fr.det[pick].box.alpha);
      taken[pick] = 1;
    }
  }
  if (!with_fp) return out;
  for (size_t j = 0; j < nd; ++j)
    if (!(taken[j] || det_flag[j] != 0 || below[j])) ++out.fp;
  int stuff = 0;
  for (const Box& dc : dontcare)
    for (size_t j = 0; j < nd; ++j) {
      if (taken[j] || det_flag[j] != 0 || below[j]) continue;
      if (overlap(fr.det[j].box, dc, 0) > kMinOverlap[cls]) {
        taken[j] = 1;
        ++stuff;
      }
    }
  out.fp -= stuff;
  if (with_aos) {
    if (out.tp > 0 || out.fp > 0) {
      double s = 0.0;                       // FPs contribute 0
      for (double d : delta) s += (1.0 + cos(d)) / 2.0;
      out.similarity = s;
    } else {
      out.similarity = -1;
    }
  }
  return out;
}

std::vector<double> recall_thresholds(std::vector<double> v, double n_gt) {
  std::sort(v.begin(), v.end(), [](double a, double b) { return a > b; });
  std::vector<double> t;
  double current = 0;
  for (size_t i = 0; i < v.size(); ++i) {
    const double left = (double)(i + 1) / n_gt;
    const double right = i + 1 < v.size() ? (double)(i + 2) / n_gt : left;
    if ((right - current) < (current - left) && i + 1 < v.size()) continue;
    t.push_back(v[i]);
    current += 1.0 / (kSamples - 1.0);
  }
  return t;
}

void evaluate(int cls, int level, const std::vector<Frame>& frames, bool with_aos, double* precision, double* aos) {
  const size_t nf = frames.size();
  std::vector<std::vector<int>> gt_flag(nf), det_flag(nf);
  std::vector<std::vector<Box>> dontcare(nf);
  std::vector<double> scores;
  int n_gt = 0;
  for (size_t f = 0; f < nf; ++f) {
    classify(cls, level, frames[f], gt_flag[f], det_flag[f], dontcare[f], n_gt);
    const Counts c = match(cls, frames[f], dontcare[f], gt_flag[f], det_flag[f], false, false, 0);
    scores.insert(scores.end(), c.tp_scores.begin(), c.tp_scores.end());
  }
  const std::vector<double> thr = recall_thresholds(scores, n_gt);
  std::vector<Counts> pr(thr.size());
  for (size_t f = 0; f < nf; ++f)
    for (size_t t = 0; t < thr.size(); ++t) {
      const Counts c = match(cls, frames[f], dontcare[f], gt_flag[f], det_flag[f], true, with_aos, thr[t]);
      pr[t].tp += c.tp;
      pr[t].fp += c.fp;
      pr[t].fn += c.fn;
      if (c.similarity != -1) pr[t].similarity += c.similarity;
    }
  for (int i = 0; i < kSamples; ++i) precision[i] = aos[i] = 0;
  for (size_t i = 0; i < thr.size() && i < (size_t)kSamples; ++i) {
    precision[i] = pr[i].tp / (double)(pr[i].tp + pr[i].fp);
    if (with_aos) aos[i] = pr[i].similarity / (double)(pr[i].tp + pr[i].fp);
  }
  for (size_t i = 0; i < thr.size() && i < (size_t)kSamples; ++i) {  // monotone from the right, over all 41 samples
    precision[i] = *std::max_element(precision + i, precision + kSamples);
    if (with_aos) aos[i] = *std::max_element(aos + i, aos + kSamples);
  }
}

}  // namespace

// precision / aos: [3 classes][3 difficulty levels][41 recall samples] doubles.
// evaluated[c] = 1 when class c has at least one detection (with x1 >= 0);
// *aos_valid = 0 when any detection carries alpha == -10 (AOS rows stay 0).
// Returns 0, or -1 (bad argument), -2 (a result file has no ground-truth file),
// -3 (result_dir/data cannot be read or is empty).
extern "C" int egn_kitti_eval_image(const char* gt_dir, const char* result_dir, int* n_frames, int* evaluated,
                                    int* aos_valid, double* precision, double* aos) {
  if (!gt_dir || !result_dir || !evaluated || !aos_valid || !precision || !aos) return -1;
  const std::string data_dir = std::string(result_dir) + "/data/";
  std::vector<int> ids;
  if (DIR* d = opendir(data_dir.c_str())) {
    while (dirent* e = readdir(d)) {
      const std::string name = e->d_name;
      if (name.size() < 10) continue;
      ids.push_back(atoi(name.substr(name.size() - 10).c_str()));
    }
    closedir(d);
  }
  if (ids.empty()) return -3;
  std::sort(ids.begin(), ids.end());
  std::vector<Frame> frames(ids.size());
  bool with_aos = true;
  for (int c = 0; c < kClasses; ++c) evaluated[c] = 0;
  for (size_t f = 0; f < ids.size(); ++f) {
    char name[32];
    snprintf(name, sizeof name, "%06d.txt", ids[f]);
    std::vector<std::vector<std::string>> rows;
    if (!read_lines(std::string(gt_dir) + "/" + name, rows)) return -2;
    for (const auto& r : rows) {
      if (r.size() < 15) continue;
      Truth g;
      g.box = Box{r[0], atof(r[4].c_str()), atof(r[5].c_str()), atof(r[6].c_str()), atof(r[7].c_str()), atof(r[3].c_str())};
      g.truncation = atof(r[1].c_str());
      g.occlusion = atoi(r[2].c_str());
      frames[f].gt.push_back(g);
    }
    rows.clear();
    if (!read_lines(data_dir + name, rows)) return -3;
    for (const auto& r : rows) {
      if (r.size() < 16) continue;
      Det d;
      d.box = Box{r[0], atof(r[4].c_str()), atof(r[5].c_str()), atof(r[6].c_str()), atof(r[7].c_str()), atof(r[3].c_str())};
      d.score = atof(r[15].c_str());
      frames[f].det.push_back(d);
      if (d.box.alpha == -10) with_aos = false;
      for (int c = 0; c < kClasses; ++c)
        if (same(d.box.type, kClassName[c])) {
          if (d.box.x1 >= 0) evaluated[c] = 1;
          break;
        }
    }
  }
  if (n_frames) *n_frames = (int)frames.size();
  *aos_valid = with_aos ? 1 : 0;
  for (int c = 0; c < kClasses; ++c)
    for (int l = 0; l < kLevels; ++l) {
      double* p = precision + (c * kLevels + l) * kSamples;
      double* a = aos + (c * kLevels + l) * kSamples;
      for (int i = 0; i < kSamples; ++i) p[i] = a[i] = 0;
      if (evaluated[c]) evaluate(c, l, frames, with_aos, p, a);
    }
  return 0;
}
