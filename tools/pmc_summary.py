"""Developer helper: mean of each PMC counter per kernel from rocprofv3 --pmc output
directories (counter_collection.csv files) given on the command line."""
import csv
import glob
import sys
from collections import defaultdict

for root in sys.argv[1:]:
    for path in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
        sums = defaultdict(lambda: [0.0, 0])
        with open(path) as stream:
            for row in csv.DictReader(stream):
                key = (row['Kernel_Name'][:96], row['Counter_Name'])
                sums[key][0] += float(row['Counter_Value'])
                sums[key][1] += 1
        for (kernel, counter), (total, n) in sorted(sums.items()):
            print('%-98s %-24s n=%4d mean=%.5g' % (kernel, counter, n, total / n))
