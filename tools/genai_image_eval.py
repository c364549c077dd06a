#!/usr/bin/env python3
"""GenAI-Bench (image) evaluation with the MI355X scorer -- command-line counterpart of the reference's
genai_image_eval.py (:9-30 arguments, :109-168 flow).  The dataset must already be on disk (no network)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import t2v_metrics_amd as t2v
from t2v_metrics_amd.genai_bench import GenAIBenchImage, run_genai_image_eval


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root_dir", default="./datasets")
    ap.add_argument("--cache_dir", default=t2v.constants.HF_CACHE_DIR)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--batch_size", default=256, type=int)
    ap.add_argument("--model", default="clip-flant5-xxl")
    ap.add_argument("--question", default=None)
    ap.add_argument("--answer", default=None)
    ap.add_argument("--result_dir", default="./genai_image_results")
    ap.add_argument("--num_prompts", default=1600, type=int, choices=[527, 1600])
    args = ap.parse_args()
    dataset = GenAIBenchImage(root_dir=args.root_dir, num_prompts=args.num_prompts)
    score_func = t2v.get_score_model(model=args.model, device=args.device, cache_dir=args.cache_dir)
    kwargs = {}
    if args.question is not None:
        kwargs["question_template"] = args.question
    if args.answer is not None:
        kwargs["answer_template"] = args.answer
    res = run_genai_image_eval(score_func, dataset, args.result_dir, args.model, batch_size=args.batch_size,
                               num_prompts=args.num_prompts, **kwargs)
    print(json.dumps(res["alignment"], indent=1))


if __name__ == "__main__":
    main()
