"""Drop-in for the score-matrix half of the reference's eval/combine_chunked_computations.py (combine_m2e_eval_results,
:143-243): concatenate per-chunk mention x entity score pickles into the one file the eval entry points read.  The
reference hard-codes its file lists and asks before overwriting; here both are arguments.  The entry points can also skip
the combined file altogether (`--score_chunks` of run_retrieval_eval_wrt_exact_crossenc.py -> anncur_amd.ingest)."""
import argparse
import logging
import sys

from anncur_amd.ingest import combine_score_chunks
from utils.zeshel_utils import N_ENTS_ZESHEL, score_matrix_filename

logging.basicConfig(stream=sys.stderr, format="%(asctime)s - %(levelname)s - %(name)s - %(message)s ", datefmt="%d/%m/%Y %H:%M:%S", level=logging.INFO)
LOGGER = logging.getLogger(__name__)


def combine_m2e_eval_results(file_list, res_dir=None, dataset_name=None, out_file=None, overwrite=False):
	"""Writes {res_dir}/{dataset_name}/ment_to_ent_scores_n_m_{total}_n_e_{N}_all_layers_False.pkl (the reference's name, :226)
	unless out_file is given."""
	if out_file is None:
		from anncur_amd.ingest import chunk_row_counts
		if res_dir is None or dataset_name not in N_ENTS_ZESHEL:
			raise ValueError("combine_m2e_eval_results: give out_file, or res_dir and a ZeShEL dataset_name")
		out_file = score_matrix_filename(res_dir, dataset_name, sum(chunk_row_counts(file_list)))
	LOGGER.info(f"Writing result to file : {out_file}")
	return combine_score_chunks(file_list, out_file, overwrite=overwrite)


def main(argv=None):
	ap = argparse.ArgumentParser(description="Combine chunked mention x entity cross-encoder score files")
	ap.add_argument("--files", nargs="+", required=True, help="chunk pickles in mention order")
	ap.add_argument("--out", type=str, default=None)
	ap.add_argument("--res_dir", type=str, default=None)
	ap.add_argument("--data_name", type=str, default=None)
	ap.add_argument("--overwrite", type=int, default=0, choices=[0, 1])
	a = ap.parse_args(argv)
	print(combine_m2e_eval_results(a.files, a.res_dir, a.data_name, a.out, bool(a.overwrite)))


if __name__ == "__main__":
	main()
