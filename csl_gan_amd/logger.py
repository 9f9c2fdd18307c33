"""Running-sum logger (reference logger.py:3-38): stats accumulate for `interval` batches, then one
line goes to stdout and one row to the CSV."""
import csv


class Logger:
    def __init__(self, str_format, stat_names, interval, csv_dir, epoch_batch_str_format="=== Epoch {} ({:2.1f}%) ===\n",
                 write_header=True):
        self.stat_names = list(stat_names)
        self.stats = dict.fromkeys(self.stat_names, 0.0)
        self.interval = interval
        self.str_format = epoch_batch_str_format + str_format
        self.f = open(csv_dir, "a")
        self.csv_writer = csv.writer(self.f)
        if write_header:
            self.csv_writer.writerow(["Epoch", "Batch"] + self.stat_names)
        self.f.flush()

    def average(self):
        for k in self.stats:
            self.stats[k] /= self.interval

    def reset_stats(self):
        for k in self.stats:
            self.stats[k] = 0.0

    def log(self, epoch, epoch_percent):
        self.average()
        row = [epoch, epoch_percent] + [self.stats[n] for n in self.stat_names]
        print(self.str_format.format(*row))
        self.csv_writer.writerow(row)
        self.f.flush()
        self.reset_stats()

    def close(self):
        self.f.close()
